"""densephrases_amd -- MI355X-native phrase-retrieval hot path for DensePhrases (MIPS.search).

Importing the package loads libdph.so (hand-written HIP for gfx950); there is no CPU fallback."""
from ._lib import DphError, Shard, merge_topk_dev  # noqa: F401
from .dump import DocMeta, DocStore  # noqa: F401
from .index import MIPS  # noqa: F401
