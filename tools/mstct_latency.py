import sys, types, torch, time
sys.path.insert(0, ".")
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.temporal_mstct import VideoNas
from computervision_codes_amd.graph import GraphedForward
args = types.SimpleNamespace(loss_type="ivt")
inter = [256, 384, 576, 864]
m = VideoNas(args, inter, 2, 8, 8, 2048, 512).eval()
m.load_state_dict(synth.fill_from_shapes(shapes.mstct_shapes(2048, inter, 2, 8, 512, "ivt"), seed=3))
x = synth.synthetic_features(256, 2048, seed=3).cuda()
g = GraphedForward(lambda xx: m.forward_btd(xx), [x])
for _ in range(5): g(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): g(x)
torch.cuda.synchronize(); print("mstct window ms", round((time.perf_counter() - t0) / 100 * 1e3, 4))
