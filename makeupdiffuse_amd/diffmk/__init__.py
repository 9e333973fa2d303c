"""Host-side mirror of the reference's ``diffmk`` package for the DDIM sampling hot path."""
