"""Checkpoint loading (drop-in for the reference's utils/models.py:10-66): strips the Lightning
``model.`` prefix, optional part-prefix filter.  Host-side file I/O; key convention is part of
the drop-in boundary."""

import logging

import torch

logger = logging.getLogger(__name__)


def load_weights_from_checkpoint(model, checkpoint_path, load_parts=None, map_location=None):
    logger.info("Loading weights from checkpoint: %s", checkpoint_path)
    checkpoint = torch.load(checkpoint_path, map_location=map_location)
    state_dict = checkpoint.get("state_dict", checkpoint)
    state_dict = {k.removeprefix("model."): v for k, v in state_dict.items()}
    if load_parts is not None:
        if isinstance(load_parts, str):
            load_parts = [load_parts]
        filtered = {k: v for k, v in state_dict.items() if any(k.startswith(f"{p}.") for p in load_parts)}
        result = model.load_state_dict(filtered, strict=False)
        for part in load_parts:
            n = sum(k.startswith(f"{part}.") for k in filtered)
            logger.info("  - %s: %s", part, f"{n} parameters loaded" if n else "NO PARAMETERS FOUND")
        return result
    model.load_state_dict(state_dict)
    return None
