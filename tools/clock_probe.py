"""Samples the GPU's shader clock and socket power (rocm-smi) while one kernel class runs back to back for a few seconds, with random
and with all-zero operands: shows whether a kernel's achieved rate sits under a power-limited clock rather than under stalls.

    python tools/clock_probe.py [seconds]      -> table on stdout (committed as profiles/<round>/clock_probe.txt)
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtl_amd
from mtl_amd import _lib

L = _lib.lib()
dev = 'cuda'
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        sclk = next((v for k, v in card.items() if 'sclk' in k.lower()), '?')
        power = next((v for k, v in card.items() if 'power' in k.lower() and 'W' in k), '?')
        return str(sclk), str(power)
    except Exception as e:      # noqa
        return 'n/a (%s)' % type(e).__name__, 'n/a'


def sustained(name, fn, flops):
    st = torch.cuda.current_stream().cuda_stream
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    one = a.elapsed_time(b) * 1e-3
    n = int(SECS / one)
    samples = []
    a.record()
    t0 = time.time()
    for i in range(n):
        fn()
        if i % max(n // 6, 1) == n // 12:
            # the host runs ahead of the device by at most the queue depth; sampling here is during the sustained load
            samples.append(smi())
    b.record()
    torch.cuda.synchronize()
    dt = a.elapsed_time(b) * 1e-3
    print('%-34s first launch %.1f TF | sustained %5.2f s: %.1f TF | sclk/power samples: %s'
          % (name, flops / one / 1e12, dt, flops * n / dt / 1e12, '  '.join('%s %sW' % s for s in samples)))


def main():
    print('idle:', smi())
    B, T, F, cin, cout = 8, 1000, 161, 64, 64
    st = lambda: torch.cuda.current_stream().cuda_stream
    for zero in (False, True):
        x = torch.relu(torch.randn(B, T, F, cin, device=dev))
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        if zero:
            x.zero_(), w.zero_()
        bias = torch.randn(cout, device=dev) * 0.1
        nb = L.mtl_conv3x3_wprep_h2_bytes(cout, cin)
        w2f, w2d = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
        y = torch.empty(B, T // 2, F // 2, cout, device=dev)
        am = torch.empty(B, T // 2, F // 2, cout, dtype=torch.uint8, device=dev)
        flops = 2.0 * B * T * F * 9 * cin * cout
        assert L.mtl_conv3x3_wprep_h2(st(), w.data_ptr(), w2f.data_ptr(), w2d.data_ptr(), cout, cin) == 0
        ax = x.abs().max().clamp_min(1e-30).reshape(1).repeat(2048)
        slot = torch.zeros(2048, device=dev)
        fn = lambda: L.mtl_conv3x3_relu_pool_fwd_h2(st(), x.data_ptr(), ax.data_ptr(), w2f.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                     am.data_ptr(), slot.data_ptr(), B, T, F, cin, cout)
        assert fn() == 0
        sustained('conv2 fwd+pool h2 (%s operands)' % ('zero' if zero else 'random'), fn, flops)
    # HBM-bound reference: a device copy
    src, dst = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
    sustained('copy 1 GiB (flops field = bytes)', lambda: dst.copy_(src), 2.0 * src.numel() * 4)


main()
