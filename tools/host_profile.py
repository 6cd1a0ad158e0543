"""cProfile of the host side of steady-state training cycles (main thread; `run_backward` = time spent waiting for the
autograd worker thread to issue the backward pass)."""
import cProfile, pstats, sys, os, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench, time
_orig = time.perf_counter
state = {'pr': None, 'n': 0}
def hooked():
    # bench.main() calls perf_counter() right before and after the headline pass: start / stop the profiler there
    state['n'] += 1
    if state['n'] == 1:
        state['pr'] = cProfile.Profile(); state['pr'].enable()
    elif state['n'] == 2:
        state['pr'].disable()
    return _orig()
bench.time.perf_counter = hooked
sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-eval', '--no-kernel-timing']
bench.main()
s = io.StringIO()
st = pstats.Stats(state['pr'], stream=s)
st.sort_stats('tottime').print_stats(22)
print(s.getvalue()[:5000])
