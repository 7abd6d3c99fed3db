"""vistaocr_amd — MI355X-native (gfx950) hot path of isi-vista/VistaOCR: CnnOcrModel (CNN -> BiLSTM -> CTC),
its criterion, greedy decoder, alphabet and train() step, all computed by hand-written HIP kernels behind the
C-ABI in include/vocr.h.  No CPU fallback: importing is cheap, computing needs libvocr.so and a GPU."""
import os as _os

# The step overlaps a side stream (weight-gradient kernels) with the main stream.  HIP spreads streams over GPU_MAX_HW_QUEUES
# hardware queues (default 4); once RCCL has created its own streams, main and side can land on ONE queue and run one after the
# other (measured with a process group initialised: 20.5 ms/step against 19.6 with 8 queues).  Only a default: a value the user
# has set is kept, and it only takes effect if this import comes before the first HIP call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .alphabet import Alphabet, arabic_alphabet, english_alphabet, french_alphabet      # noqa: F401,E402
from .ctc import CTCLoss                                                                # noqa: F401,E402
from .decoder import ArgmaxDecoder                                                      # noqa: F401,E402
from .model import CnnOcrModel                                                          # noqa: F401,E402
from .train import FlatClampAdam, make_optimizer, seed_rank, train, train_async                                    # noqa: F401,E402
from .dataset import OcrDataset                                                         # noqa: F401,E402

__all__ = ["Alphabet", "english_alphabet", "arabic_alphabet", "french_alphabet", "CTCLoss", "ArgmaxDecoder",
           "CnnOcrModel", "FlatClampAdam", "make_optimizer", "seed_rank", "train", "train_async", "OcrDataset"]
