import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgm_amd.synth import make_stream
stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev)
starts = loader._starts
with hm.activate('bench'):
    prev = None
    for i in range(60):
        prev = loader(starts[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60, 260):
        prev = loader(starts[i])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host us/call %.1f, wall us/call %.1f' % (1e6 * (t1 - t0) / 200, 1e6 * (t2 - t0) / 200))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(260, 560):
        prev = loader(starts[i])
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
