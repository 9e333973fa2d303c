"""Drop-in alias: ``diffmk.makeups`` resolves to the MI355X implementation."""
from makeupdiffuse_amd.diffmk.makeups import *  # noqa: F401,F403
from makeupdiffuse_amd.diffmk.makeups import BaseModel  # noqa: F401
