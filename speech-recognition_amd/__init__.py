"""speech_recognition_amd - MI355X-native (gfx950) training hot path of cosmoquester/speech-recognition.

Mirrors the reference's module layout for the path `speech_recognition.run.train` exercises
(`data`, `measure`, `utils`, `configs`, `models`), with all arithmetic in hand-written HIP kernels
behind the C ABI of include/asr_mi355x.h (libasr_mi355x.so).  The directory is named
``speech-recognition_amd``; import it as ``speech_recognition_amd`` (alias module at the repo root).
"""
__version__ = "0.1.0"
