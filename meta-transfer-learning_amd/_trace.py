"""Host-side phase trace of an enqueued meta-iteration (diagnostics, MTL_TRACE_PHASES=1): where the HOST spends the time between
entering TransientTrainer.enqueue_iteration and returning from it -- python preparation, staging waits, the command-list replay
(with the host time of every replayed call from mtl_cmdlist_run_timed), read-back copies, the outer step.  A device profile cannot
show a launch call that blocks inside the runtime; this can.  Off: every hook is one attribute test."""
import os
import time

ON = os.environ.get('MTL_TRACE_PHASES', '0') not in ('', '0')
_marks = []
steps = []          # one dict per traced iteration: phase -> ms (+ 'slow_calls': [(ms, index, function)])
_slow = []


def begin():
    if ON:
        del _marks[:]
        del _slow[:]
        _marks.append(('begin', time.perf_counter()))


def mark(name):
    if ON:
        _marks.append((name, time.perf_counter()))


def calls(names, host_us, base):
    """host time per replayed call (microseconds): keep the slowest few and the total"""
    if ON:
        order = sorted(range(len(host_us)), key=lambda i: -host_us[i])[:4]
        _slow.extend((round(host_us[i] * 1e-3, 2), base + i, names[i]) for i in order if host_us[i] > 500.0)


def end():
    if ON and _marks:
        out, prev = {}, _marks[0][1]
        for name, t in _marks[1:]:
            out[name] = round(out.get(name, 0.0) + (t - prev) * 1e3, 2)
            prev = t
        out['total'] = round((prev - _marks[0][1]) * 1e3, 2)
        if _slow:
            out['slow_calls'] = sorted(_slow, reverse=True)[:6]
        steps.append(out)
