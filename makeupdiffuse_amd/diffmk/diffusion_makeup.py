"""``diffmk.diffusion_makeup`` — names the reference's yaml / scripts point at (diffmodels/base_diffusion_makeup.yaml:2
`BaseDoubleControlModel`; runs/test.py's experiment yaml -> `TestDoubleControlModel`, diffusion_makeup.py:308)."""
from .makeup_diffuse import BaseMakeUpDiffuse, TestDiffuseModel


class BaseDoubleControlModel(BaseMakeUpDiffuse):
    pass


class TestDoubleControlModel(TestDiffuseModel):
    pass
