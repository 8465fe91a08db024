"""MI355X-native (gfx950) VMAE predictor forward pass for Counterfactual World Models.

Only the hot path of neuroailab/CounterfactualWorldModels lives here: the masked predictor
(`vmae`), the wrapper surface directly around it (`prediction`), and the HIP library they call
(`csrc/` -> `lib/libcwm_hip.so`, C ABI in `include/cwm_hip.h`).  See DESIGN.md.
"""
from .config import CONFIGS, VmaeConfig, algorithmic_flops, num_parameters, state_dict_schema  # noqa: F401

__version__ = "0.1.0"
