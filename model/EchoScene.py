from echoscene_amd.model.scene import Sg2ScDiffModel  # noqa: F401
