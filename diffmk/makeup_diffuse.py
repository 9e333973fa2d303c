"""Drop-in alias: ``diffmk.makeup_diffuse`` resolves to the MI355X implementation."""
from makeupdiffuse_amd.diffmk.makeup_diffuse import *  # noqa: F401,F403
