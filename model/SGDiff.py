from echoscene_amd.model.scene import SGDiff  # noqa: F401
