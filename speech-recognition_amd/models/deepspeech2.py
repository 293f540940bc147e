"""DeepSpeech2 on MI355X (mirror of speech_recognition/models/deepspeech2.py) - under construction."""
from .model_proto import ModelProto


def ctc_loss_only(y_true, y_pred, blank_index, pad_index=0):
    raise NotImplementedError("CTC kernels are not built yet")


class DeepSpeech2(ModelProto):
    model_checkpoint_path = "model-{epoch}epoch-{val_loss:.4f}loss.ckpt"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("DeepSpeech2 is not built yet")
