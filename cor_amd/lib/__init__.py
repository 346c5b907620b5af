"""Mirror of the reference's `lib` package (module and attribute names, state_dict keys) on the HIP engine.

Every class below the top-level model is a PARAMETER HOLDER: it fixes the state_dict key contract and has no
arithmetic of its own (that lives in cor_amd/csrc behind cor_amd.engine). Calling one of them directly raises a clear error
instead of nn.Module's generic "missing forward"."""
import inspect as _inspect

from torch import nn as _nn

from .sam_model.common import no_standalone_forward as _no_fwd
from .sam_model import common as _c, image_encoder as _ie, mask_decoder as _md, my_prompt_encoder as _pe, transformer as _tr
from .support_model import cir_feature_fuse as _cf, mask_adapter as _ma, siglip_openclip as _so
from . import support_branch as _sb

for _m in (_c, _ie, _md, _pe, _tr, _cf, _ma, _so, _sb):
    for _name, _cls in _inspect.getmembers(_m, _inspect.isclass):
        if issubclass(_cls, _nn.Module) and _cls.__module__ == _m.__name__ and "forward" not in _cls.__dict__:
            _cls.forward = _no_fwd
