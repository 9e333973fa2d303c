"""Top-level alias package so `import diffmk.cddim` / `diffmk.makeup_diffuse` keep working."""
