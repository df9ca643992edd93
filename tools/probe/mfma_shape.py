"""32x32x16 vs 16x16x32 fp16 matrix instructions in the LDS-fed step of csrc/mtl_probe.hip (modes 1 / 4), zeros / random / post-ReLU operands.  GPU only."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
lib = ctypes.CDLL(os.path.join(R, 'meta-transfer-learning_amd', 'libmtl_probe.so'))
lib.mtl_probe_mfma_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda')
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
steps = 3000
for rep in range(2):
    for mode, name in ((1, 'LDS-fed 32x32x16'), (4, 'LDS-fed 16x16x32'), (0, 'registers 32x32x16')):
        row = []
        for fill in (0, 1, 2):
            lib.mtl_probe_mfma_f16(st, ncu, steps, mode, fill, sink.data_ptr())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record(); lib.mtl_probe_mfma_f16(st, ncu, steps, mode, fill, sink.data_ptr()); ev[1].record()
            torch.cuda.synchronize()
            row.append(ncu * steps * 8 * 24 * 32768.0 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12)
        print('%-20s zeros %7.1f TF | random %7.1f TF | post-ReLU %7.1f TF' % (name, *row))
