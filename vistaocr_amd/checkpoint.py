"""Checkpoint interop with the reference (src/train_cnn_lstm.py:427-438 writes, src/models/cnnlstm.py:40-71 and
src/utils/decode.py:58-71 read).

A reference-written .pth pickles `model_hyper_params['alphabet']` as an instance of the top-level class
`alphabet.Alphabet` (src/alphabet.py:1-12) and, when trained with multigpu=True, carries nn.DataParallel's `cnn.module.`
key prefix.  load() resolves that class name to vistaocr_amd.alphabet.Alphabet (same attributes) through the unpickler's
find_class — nothing is registered in sys.modules — and save() writes the same global name back, so files move both ways.
"""
import io
import pickle
import types

import torch

from .alphabet import Alphabet

_REF_MODULE, _REF_NAME = "alphabet", "Alphabet"


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_MODULE and name == _REF_NAME:
            return Alphabet
        return super().find_class(module, name)


class _Pickler(pickle._Pickler):
    """Pure-Python pickler (the tensor bytes go through torch's persistent ids, so the pickle stream is tiny) that names
    our Alphabet class the way the reference's own module does."""

    def save_global(self, obj, name=None):
        if obj is Alphabet:
            self.write(pickle.GLOBAL + (_REF_MODULE + "\n" + _REF_NAME + "\n").encode("ascii"))
            self.memoize(obj)
            return
        super().save_global(obj, name)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = pickle._Pickler.save_type


def _module(**overrides):
    m = types.ModuleType("vistaocr_amd._checkpoint_pickle")
    m.__dict__.update({k: v for k, v in pickle.__dict__.items() if not k.startswith("__")})
    m.__dict__.update(overrides)
    return m


def _load_fn(file, **kw):
    return _Unpickler(file, **kw).load()


def _dump_fn(obj, file, protocol=2, **kw):
    _Pickler(file, protocol=protocol).dump(obj)


_LOAD_MODULE = _module(Unpickler=_Unpickler, load=_load_fn)
_SAVE_MODULE = _module(Pickler=_Pickler, dump=_dump_fn)


def load(path, map_location="cpu"):
    """torch.load of a checkpoint written by the reference or by save()."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_LOAD_MODULE)


def save(obj, path):
    """torch.save with the reference's class path for Alphabet instances (protocol 2, like torch's default)."""
    torch.save(obj, path, pickle_module=_SAVE_MODULE, pickle_protocol=2)


def strip_dataparallel_prefix(state_dict):
    """'cnn.module.X' -> 'cnn.X' (src/utils/decode.py:58-71): this build never wraps the CNN in nn.DataParallel."""
    return {(k.replace("cnn.module.", "cnn.", 1) if k.startswith("cnn.module.") else k): v for k, v in state_dict.items()}


def add_dataparallel_prefix(state_dict):
    """'cnn.X' -> 'cnn.module.X': what the reference's default multigpu=True model expects under strict=True."""
    return {("cnn.module." + k[4:] if k.startswith("cnn.") and not k.startswith("cnn.module.") else k): v
            for k, v in state_dict.items()}
