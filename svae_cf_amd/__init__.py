"""svae_cf_amd: MI355X-native training / scoring step of the Sequential VAE for collaborative
filtering (reference: noveens/svae_cf, main_svae.ipynb).  HIP kernels behind a C-ABI
(include/svae_hip.h), Python host mirroring the notebook's Model / VAELoss / train / evaluate."""
from .batch import BatchPlan, build_plan, pack_rows, unpack_rows  # noqa: F401
from .engine import PREC_BF16, PREC_FP32, PREC_SPLIT, SVAEEngine  # noqa: F401
from .layout import ParamLayout  # noqa: F401
from . import notebook_api  # noqa: F401,E402
