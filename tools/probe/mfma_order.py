"""Does the order of the three-instruction products (which operand register consecutive matrix instructions share) change the rate the chip sustains?
csrc/mtl_probe.hip modes 0 / 1 (per accumulator: a1 b0, a0 b1, a0 b0) vs 2 / 3 (the four instructions with a0 back to back).  GPU only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'meta-transfer-learning_amd', 'libmtl_probe.so'))
lib.mtl_probe_mfma_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda')
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
steps = 3000
for rep in range(2):
    for mode, name in ((0, 'registers, per-accumulator order'), (2, 'registers, shared-operand order'), (1, 'LDS-fed, per-accumulator order'), (3, 'LDS-fed, shared-operand order'), (4, 'LDS-fed, 16x16x32 instructions')):
        for fill, fn in ((1, 'random'), (2, 'post-ReLU')):
            lib.mtl_probe_mfma_f16(st, ncu, steps, mode, fill, sink.data_ptr())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record(); lib.mtl_probe_mfma_f16(st, ncu, steps, mode, fill, sink.data_ptr()); ev[1].record()
            torch.cuda.synchronize()
            print('%-36s %-10s %7.1f TF' % (name, fn, ncu * steps * 8 * 24 * 32768.0 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12))
