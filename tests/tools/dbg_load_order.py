import sys, ctypes as C
sys.path.insert(0, '.')
import torch
mode = sys.argv[1]
from speech_recognition_amd import _lib
if mode == "cdll":
    C.CDLL(_lib.LIB_PATH)
elif mode == "load":
    _lib.load()
elif mode == "sizes":
    lib = C.CDLL(_lib.LIB_PATH)
    lib.asr_struct_size.restype = C.c_long
    print(lib.asr_struct_size(b"asr_rnn_seq"))
maps = [l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l]
print(sorted(set(maps)))
x = torch.zeros(4, device="cuda")
from speech_recognition_amd import ops
y = torch.ones(1024, device="cuda")
ops.fill(y, 3.0)
torch.cuda.synchronize()
print(mode, "ok", float(y[0]))
