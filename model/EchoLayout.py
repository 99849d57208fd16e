from echoscene_amd.model.scene import Sg2BoxDiffModel  # noqa: F401
