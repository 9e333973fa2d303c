"""Drop-in alias: ``diffmk.diffusion_makeup`` resolves to the MI355X implementation."""
from makeupdiffuse_amd.diffmk.diffusion_makeup import *  # noqa: F401,F403
