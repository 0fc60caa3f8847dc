"""Base segmentation model (drop-in for the reference's models/segmentation/base.py)."""

from torch import Tensor, nn


class BaseSegmentationModel(nn.Module):
    """Wiring + substring-match freezing (base.py:8-44)."""

    def __init__(self, encoder=None, neck=None, decoder=None, head=None, output_struct=None,
                 auxilary_head=None) -> None:
        super().__init__()
        self.encoder = encoder
        self.neck = neck
        self.decoder = decoder
        self.auxilary_head = auxilary_head
        self.head = head
        self.output_struct = output_struct

    def forward(self, x: Tensor) -> Tensor:
        x = self.encoder(x)
        x = self.neck(x)
        x = self.decoder(x)
        aux = None
        if self.auxilary_head:
            aux = self.auxilary_head(x)
        x = self.head(x)
        return self.output_struct(out=x, aux=aux)

    def _freeze_layers(self, layers: list[str]) -> None:
        for name, param in self.named_parameters():
            if any(layer in name for layer in layers):
                param.requires_grad = False
