"""Prototype structure of the ASR models (mirror of speech_recognition/models/model_proto.py:7-54).

The reference's ModelProto is a tf.keras.Model; here it is a plain Python object whose arithmetic
runs in libasr_mi355x.so.  The members run.train relies on are kept: ``__call__(inputs, training)``,
``get_loss_fn()``, ``get_metrics()``, static ``get_batching_shape`` / ``make_example`` and the class
attribute ``model_checkpoint_path``; ``summary()`` / ``load_weights`` / ``save_weights`` stand in for
the Keras methods of the same names.
"""
from abc import ABCMeta, abstractmethod
from typing import Callable, List, Optional

import torch


class ModelProto(metaclass=ABCMeta):
    model_checkpoint_path: str = ""

    def __init__(self, *args, **kwargs):
        self.built = False
        self.store = None          # ParamStore with the trainable variables
        self.buffers = {}          # non-trainable variables (BatchNormalization moving statistics)

    def __call__(self, inputs, training: Optional[bool] = None):
        return self.call(inputs, training=training)

    @abstractmethod
    def call(self, inputs, training: Optional[bool] = None) -> torch.Tensor:
        pass

    @abstractmethod
    def get_loss_fn(self) -> Callable:
        pass

    @abstractmethod
    def get_metrics(self) -> List:
        pass

    @staticmethod
    @abstractmethod
    def get_batching_shape(audio_pad_length: Optional[int], token_pad_length: Optional[int], frequency_dim: int,
                           feature_dim: int):
        """Return shapes of padded batch (model_proto.py:30-42)."""

    @staticmethod
    @abstractmethod
    def make_example(audio, tokens):
        """Make training example (MODEL_INPUT, Y_TRUE) from audio input and token output."""

    def raise_on_sweep_timeout(self):
        """Forward-only passes (search, evaluation, model(...)) run the one-launch sweeps too; their time-out flag is only folded into
        the sticky error word by a training update.  Call this where the host synchronises anyway (it does): raises, and clears the
        flag, if a hand-off timed out since the flag was last cleared - the outputs of those passes are invalid."""
        flag = getattr(self.store, "err_flag", None) if self.store is not None else None
        if flag is not None and float(flag[0]) != 0.0:
            flag.zero_()
            raise RuntimeError("one-launch recurrent sweep: an inter-workgroup hand-off timed out during a forward pass; its outputs are "
                               "invalid (rerun with ASR_PERSISTENT_RNN=0 to use the per-step kernels)")

    # ---- Keras-method stand-ins used by run.train -------------------------------------------------
    def count_params(self):
        n = self.store.num_trainable() if self.store is not None else 0
        return n + sum(v.numel() for v in self.buffers.values())

    def summary(self, print_fn=print):
        print_fn(f'Model: "{type(self).__name__}"')
        if self.store is None:
            print_fn("  (not built)")
            return
        for n, s in self.store.shapes.items():
            print_fn(f"  {n:70s} {str(tuple(s)):>20s}")
        for n, v in self.buffers.items():
            print_fn(f"  {n:70s} {str(tuple(v.shape)):>20s}  (non-trainable)")
        print_fn(f"Total params: {self.count_params():,}  (trainable {self.store.num_trainable():,})")

    def state_dict(self):
        d = self.store.state_dict()
        d.update({k: v.detach().cpu().clone() for k, v in self.buffers.items()})
        return d

    def load_state_dict(self, values):
        self.store.load(values)
        for k, v in values.items():
            if k in self.buffers:
                self.buffers[k].copy_(torch.as_tensor(v).to(torch.float32))
        self.weights_changed()

    def save_weights(self, path):
        """Keras `save_weights(path)` (run/train.py:208-212): a TensorFlow tensor-bundle checkpoint
        (<path>.index + <path>.data-00000-of-00001) with the reference's variable paths, so the files are
        interchangeable with the reference's.  A path ending in '.pt' writes a torch state dict instead."""
        if path.endswith(".pt"):
            torch.save(self.state_dict(), path)
            return
        from ..checkpoint import save_variables
        save_variables(path, {k: v.numpy() for k, v in self.state_dict().items()})

    def load_weights(self, path):
        """Keras `load_weights(path)` (run/train.py:152-154): reads a tensor-bundle checkpoint written by the
        reference or by save_weights; every model variable must be present with its exact shape."""
        import os
        if path.endswith(".pt") or (os.path.isfile(path) and not os.path.exists(path + ".index")):
            self.load_state_dict(torch.load(path, map_location="cpu"))
            return
        from ..checkpoint import load_variables
        values = load_variables(path)
        mine = self.state_dict()
        missing = sorted(set(mine) - set(values))
        if missing:
            raise ValueError(f"load_weights: {path} lacks {len(missing)} variables of {type(self).__name__}, e.g. {missing[:3]}")
        for k, v in mine.items():
            if tuple(values[k].shape) != tuple(v.shape):
                raise ValueError(f"load_weights: {k} has shape {tuple(values[k].shape)} in {path}, the model needs {tuple(v.shape)}")
        self.load_state_dict({k: torch.from_numpy(values[k]) for k in mine})

    def release_workspace(self, ws):
        """Forget a workspace handed out by train_workspace()/forward() so that its buffers can be freed."""
        cache = getattr(self, "_ws", None)
        if cache:
            for k in [k for k, v in cache.items() if v is ws]:
                del cache[k]

    def weights_changed(self):
        """Re-derive any packed weight image (call after every optimizer step / weight load)."""
