"""CPU oracle for the speech_recognition.run.train hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path (``speech-recognition_amd/``)
never imports this package and fails loudly when the HIP library is missing.

What it is: a restatement, in numpy / torch-CPU (float64 by default), of the
algorithms that cosmoquester/speech-recognition delegates to TensorFlow 2 / Keras on
the training path.  Every function cites the reference ``file:line`` it follows;
TensorFlow-internal semantics that are not in the reference tree are marked [TF-sem]
(public TF 2.4/2.5 behaviour; TensorFlow is not installed here and cannot be run).

PINNING STATUS
  * pinned by reference fixtures: log-mel of silence == log(1e-12) = -27.631021 with
    shape [412, 80, 1] (tests/data/wav_dataset.tfrecord, reference tests/test_data.py:53-57);
    frame-count formula; SpecAugment bounds; mask-padding invariance of BiRNN/Recurrent;
    output shapes; Keras weight layouts (shapes in tests/data/model-checkpoints/*.ckpt).
  * everything else (STFT/mel arithmetic on non-silent audio, every model output, loss,
    gradient and optimizer value): **parity unpinned** - no reference test or golden
    vector holds a number for it and the TF reference cannot be executed in this image.
    The oracle is cross-checked against an independent second route (torch.nn.LSTM/GRU,
    F.conv2d, torch.stft, F.ctc_loss) in tests/test_oracle.py instead.
"""
