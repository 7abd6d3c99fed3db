"""vistaocr_amd — MI355X-native (gfx950) hot path of isi-vista/VistaOCR: CnnOcrModel (CNN -> BiLSTM -> CTC),
its criterion, greedy decoder, alphabet and train() step, all computed by hand-written HIP kernels behind the
C-ABI in include/vocr.h.  No CPU fallback: importing is cheap, computing needs libvocr.so and a GPU."""
from .alphabet import Alphabet, arabic_alphabet, english_alphabet, french_alphabet      # noqa: F401
from .ctc import CTCLoss                                                                # noqa: F401
from .decoder import ArgmaxDecoder                                                      # noqa: F401
from .model import CnnOcrModel                                                          # noqa: F401
from .train import FlatClampAdam, make_optimizer, train, train_async                                    # noqa: F401
from .dataset import OcrDataset                                                         # noqa: F401

__all__ = ["Alphabet", "english_alphabet", "arabic_alphabet", "french_alphabet", "CTCLoss", "ArgmaxDecoder",
           "CnnOcrModel", "FlatClampAdam", "make_optimizer", "train", "train_async", "OcrDataset"]
