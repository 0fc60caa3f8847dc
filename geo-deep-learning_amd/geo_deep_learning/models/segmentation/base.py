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


class EncoderMixin:
    """Encoder mixin (base.py:47-73): channel bookkeeping shared with smp-style encoders."""

    _output_stride = 32

    @property
    def out_channels(self) -> list[int]:
        return self._out_channels[: self._depth + 1]

    @property
    def output_stride(self) -> int:
        return min(self._output_stride, 2**self._depth)

    def set_in_channels(self, in_channels: int, *, pretrained: bool = True) -> None:  # noqa: ARG002
        if in_channels == 3:
            return
        self._in_channels = in_channels
        if self._out_channels[0] == 3:
            self._out_channels = (in_channels, *self._out_channels[1:])
