"""Drop-in alias: ``diffmk.makeup_controlnet`` resolves to the MI355X implementation."""
from makeupdiffuse_amd.diffmk.makeup_controlnet import *  # noqa: F401,F403
