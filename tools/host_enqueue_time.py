import time, torch, sys
sys.path.insert(0, '/root/repo')
import yolo_amd
from yolo_amd import engine
from yolo_amd.synthetic import images, init_deterministic
from yolo_amd.yolov5_official.model import Model
from yolo_amd.yolov5_official.utils import non_max_suppression_padded
dev = torch.device('cuda', 0)
engine.set_compute_dtype('bf16')
m = Model(); init_deterministic(m, seed=0, head_convs=list(m.model[-1].m)); m = m.to(dev).eval()
x = images(64, 640, seed=100).to(dev)
with torch.no_grad():
    for _ in range(5):
        z, _ = m(x); non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        z, _ = m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(20):
        z, _ = m(x); non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
print(f"forward enqueue {(t1-t0)/20*1e3:.3f} ms/step (gpu total {(t2-t0)/20*1e3:.3f}); fwd+nms enqueue {(t3-t2)/20*1e3:.3f} ms/step (total {(t4-t2)/20*1e3:.3f})")
