"""CPU oracle for the MakeupDiffuse DDIM sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``makeupdiffuse_amd/`` (the product)
imports this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED at the third-party boundary: the arithmetic of the UNet /
ControlNet / stock DDIM sampler lives in the un-vendored ``ldm`` / ``cldm``
packages (lllyasviel/ControlNet, version unpinned by the reference), the
reference ships no golden vectors, and ``ldm``/``cldm`` are not importable
here (ordinary ModuleNotFoundError).  What pins this oracle instead:
  * the reference's own lines it restates (cited per function),
  * the DDIM schedule known-answer values of SURVEY.md App. B,
  * the parameter counts 859.5 M (UNet) / 361.3 M (ControlNet),
  * algebraic identities (zero zero-convs == control-free UNet, CFG s=1, ...).
"""
