import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from eioku_amd import scene, synth, _lib
_lib.init(0); gpu = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
f = synth.frames_bgr(1234, 64, h, w, gpu)
y = f[..., 1].contiguous()
for _ in range(3):
    scene.hsv_sums(f, keep_on_device=True); scene.luma_sad(y, keep_on_device=True)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): scene.hsv_sums(f, keep_on_device=True)
torch.cuda.synchronize(); dt = (time.time() - t) / 10
print('hsv ms', dt * 1e3, 'GB/s', 64 * h * w * 3 / dt / 1e9)
t = time.time()
for _ in range(10): scene.luma_sad(y, keep_on_device=True)
torch.cuda.synchronize(); dt = (time.time() - t) / 10
print('sad ms', dt * 1e3, 'GB/s', 64 * h * w / dt / 1e9)
