import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, types
import train_eval_scene as T
from nerfstudio_thermal_amd.pipeline import ThermalPipeline
from nerfstudio_thermal_amd.dataparser import load_image_float32
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tmp = tempfile.TemporaryDirectory()
T.write_cube_scene(tmp.name, 12, dev)
pipe = ThermalPipeline(tmp.name, device=dev)
pipe.train(3000)
print("eval split:", pipe.get_average_eval_image_metrics())
# a TRAINING view rendered through the eval path
m = pipe.model; m.eval()
tr = pipe.train_outputs
for i in (0, 5, 11, 16):
    c = tr.cameras
    cam = types.SimpleNamespace(camera_to_worlds=c["c2w"][i], fx=float(c["fx"][i]), fy=float(c["fy"][i]), cx=float(c["cx"][i]), cy=float(c["cy"][i]),
                                width=int(c["width"][i]), height=int(c["height"][i]), distortion_params=c["distortion"][i], camera_index=i)
    outs = m.get_outputs_for_camera(cam)
    gt = load_image_float32(tr.image_filenames[i]).to(dev)
    met, _ = m.get_image_metrics_and_images(outs, {"image": gt, "is_thermal": int(tr.metadata["is_thermal"][i])})
    print("train view", i, tr.metadata["is_thermal"][i], met, "acc mean", float(outs["accumulation"].mean()))
print("pose adjustment norms", m.arena.view("camera_optimizer.pose_adjustment").norm(dim=1))
