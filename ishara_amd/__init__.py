"""ishara_amd — MI355X-native (gfx950) hot path of the Ishara ASL-fingerspelling recogniser:
the Conv1D -> Squeezeformer -> Conformer CTC encoder behind the reference's get_model(...) /
model.fit surface.  See DESIGN.md and include/ishara_hip.h."""
from .model import (Callback, History, LearningRateScheduler, Model, Optimizer, PAD_TOKEN_IDX,  # noqa: F401
                    get_model, lrfn, make_config)
from ._lib import IsharaError  # noqa: F401
from .conformer import ConformerEncoder  # noqa: F401  (torch family: conformer/conformer.py)
from .squeezeformer import Squeezeformer, SqueezeformerEncoder  # noqa: F401  (torch family: squeezeformer/encoder.py, model.py)
