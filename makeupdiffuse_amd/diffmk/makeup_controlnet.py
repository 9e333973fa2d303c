"""``diffmk.makeup_controlnet`` — the earlier ControlNet variant's cond contract (reference
diffmk/makeup_controlnet.py:137-167): a 6-channel NCHW hint, source first, and c_crossattn = [txt]."""
from .cddim import MKDDIMSampler  # noqa: F401  (the reference module re-exports it, :5)
from .makeup_diffuse import BaseMakeUpDiffuse


class MakeupDoubleControlModel(BaseMakeUpDiffuse):
    pass
