"""drop-in module tree: same import paths as the reference, so its Hydra `target:` strings resolve here."""
