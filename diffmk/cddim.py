"""Drop-in alias: ``diffmk.cddim`` resolves to the MI355X implementation."""
from makeupdiffuse_amd.diffmk.cddim import *  # noqa: F401,F403
