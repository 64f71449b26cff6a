"""N4: forward Gaussian-splat render with an RGB + thermal colour per Gaussian ("thermal-splatfacto", BASELINE config 4).

The reference has no thermal-splatfacto method; the boundary mirrored here is stock `SplatfactoModel.get_outputs(camera)`
(nerfstudio/models/splatfacto.py:659-822) with the parameter names of its `gauss_params` (means, scales, quats, opacities, features_dc,
features_rest) plus a second set of SH coefficients with one channel (features_dc_thermal / features_rest_thermal) rendered through the
same rasteriser -- the splat analogue of thermal-nerfacto's shared density.  The three gsplat calls (project_gaussians,
spherical_harmonics, rasterize_gaussians x2) run as tn_splat_project / tn_splat_bin / tn_splat_raster of libthermal_nerf_hip.so.
Forward only (eval render): densification, the SSIM loss and the backward pass are out of scope.  Parity is unpinned (gsplat is a third-party
package outside the reference tree; oracle/splat_oracle.py restates its published algorithm).  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from . import _lib
from .ops import _stream

BLOCK_WIDTH = 16  # splatfacto.py:738


@dataclass
class ThermalSplatfactoModelConfig:
    """The fields of SplatfactoModelConfig (splatfacto.py:103-172) that the forward render reads."""

    sh_degree: int = 3
    sh_degree_interval: int = 1000
    rasterize_mode: str = "classic"  # or "antialiased"
    background_color: str = "black"  # "black" | "white" (eval render; "random" is a training-only setting)
    background_thermal: float = 0.0
    num_random: int = 50000
    random_scale: float = 10.0


@dataclass
class PinholeCamera:
    """One perspective camera: what SplatfactoModel.get_outputs reads from `Cameras` (camera_to_worlds [3,4] in nerfstudio's convention --
    x right, y up, z back -- and the intrinsics)."""

    camera_to_world: Tensor
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> Tensor:
    """splatfacto.py:82-100."""
    t = znear * math.tan(0.5 * fovy)
    b = -t
    r = znear * math.tan(0.5 * fovx)
    l = -r  # noqa: E741
    n, f = znear, zfar
    return torch.tensor([[2 * n / (r - l), 0.0, (r + l) / (r - l), 0.0], [0.0, 2 * n / (t - b), (t + b) / (t - b), 0.0],
                         [0.0, 0.0, (f + n) / (f - n), -1.0 * f * n / (f - n)], [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32)


def camera_struct(cam: PinholeCamera, clip_thresh: float = 0.01) -> _lib.TnSplatCamera:
    """splatfacto.py:700-720: flip y/z into gsplat's convention, invert analytically, build the full projection matrix (host side, 4x4)."""
    c2w = cam.camera_to_world.detach().float().cpu()
    R = c2w[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    T = c2w[:3, 3:4]
    R_inv = R.T
    T_inv = -R_inv @ T
    viewmat = torch.eye(4)
    viewmat[:3, :3] = R_inv
    viewmat[:3, 3:4] = T_inv
    fovx = 2 * math.atan(cam.width / (2 * cam.fx))
    fovy = 2 * math.atan(cam.height / (2 * cam.fy))
    proj = projection_matrix(0.001, 1000, fovx, fovy) @ viewmat
    s = _lib.TnSplatCamera()
    for i, v in enumerate(viewmat[:3].reshape(-1).tolist()):
        s.viewmat[i] = v
    for i, v in enumerate(proj.reshape(-1).tolist()):
        s.projmat[i] = v
    s.fx, s.fy, s.cx, s.cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
    for i, v in enumerate(c2w[:3, 3].tolist()):
        s.position[i] = v
    s.clip_thresh = clip_thresh
    s.width, s.height = int(cam.width), int(cam.height)
    return s


def _ptr(t: Optional[Tensor], dtype, name: str):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous {dtype} HIP tensor (the splat path has no CPU fallback)")
    return C.c_void_p(t.data_ptr())


class ThermalSplatfactoModel(nn.Module):
    """Forward render of RGB + thermal Gaussians.  `gauss_params` keeps the reference's names (splatfacto.py:226-235)."""

    def __init__(self, config: Optional[ThermalSplatfactoModelConfig] = None, num_points: Optional[int] = None, device="cuda", seed: int = 0):
        super().__init__()
        self.config = config or ThermalSplatfactoModelConfig()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ThermalSplatfactoModel needs a HIP device: there is no CPU fallback on this path")
        _lib.load()
        n = self.config.num_random if num_points is None else num_points
        g = torch.Generator().manual_seed(seed)
        dim_sh = (self.config.sh_degree + 1) ** 2
        # random_init of the reference (splatfacto.py:190-225): positions uniform in a cube, identity-ish colours, opacity logit(0.1)
        means = (torch.rand((n, 3), generator=g) - 0.5) * self.config.random_scale
        self.gauss_params = nn.ParameterDict({
            "means": nn.Parameter(means.to(dev)),
            "scales": nn.Parameter(torch.full((n, 3), math.log(0.01 * self.config.random_scale)).to(dev)),
            "quats": nn.Parameter(torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=-1).to(dev)),
            "opacities": nn.Parameter(torch.logit(0.1 * torch.ones(n, 1)).to(dev)),
            "features_dc": nn.Parameter(torch.rand((n, 3), generator=g).to(dev)),
            "features_rest": nn.Parameter(torch.zeros((n, dim_sh - 1, 3), device=dev)),
            "features_dc_thermal": nn.Parameter(torch.rand((n, 1), generator=g).to(dev)),
            "features_rest_thermal": nn.Parameter(torch.zeros((n, dim_sh - 1, 1), device=dev)),
        })
        self.step = 0
        self._ws: Optional[Tensor] = None
        self._cap = 0
        self.last_projection: Dict[str, Tensor] = {}
        self.last_num_intersections = 0

    # the reference's accessors
    @property
    def num_points(self) -> int:
        return self.gauss_params["means"].shape[0]

    @property
    def means(self):
        return self.gauss_params["means"]

    def load_gaussians(self, params: Dict[str, Tensor]) -> None:
        dev = self.means.device
        self.gauss_params = nn.ParameterDict({k: nn.Parameter(v.detach().float().contiguous().to(dev)) for k, v in params.items()})

    def _workspace(self, n: int, cap: int, tiles: int) -> Tensor:
        need = int(_lib.load().tn_splat_workspace_bytes(n, cap, tiles))
        if need < 0:
            raise RuntimeError("tn_splat_workspace_bytes: bad sizes")
        if self._ws is None or self._ws.numel() < need or self._cap != cap:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.means.device)
            self._cap = cap
        return self._ws

    @torch.no_grad()
    def get_outputs(self, camera: PinholeCamera) -> Dict[str, Tensor]:
        """splatfacto.py:659-822 (eval mode, no crop box): project -> SH colours -> tile binning -> raster (colour + depth in one pass).
        Returns rgb [H,W,3], thermal [H,W,1], depth [H,W,1], accumulation [H,W,1], background [3]."""
        cfg = self.config
        lib = _lib.load()
        gp = self.gauss_params
        dev = gp["means"].device
        N, H, W = self.num_points, int(camera.height), int(camera.width)
        if cfg.rasterize_mode not in ("classic", "antialiased"):
            raise ValueError(f"Unknown rasterize_mode: {cfg.rasterize_mode}")
        aa = int(cfg.rasterize_mode == "antialiased")
        bg = torch.ones(3) if cfg.background_color == "white" else torch.zeros(3)
        cam = camera_struct(camera)
        tiles = ((W + BLOCK_WIDTH - 1) // BLOCK_WIDTH) * ((H + BLOCK_WIDTH - 1) // BLOCK_WIDTH)
        K = gp["features_rest"].shape[1]
        # degree evaluated at this step (splatfacto.py:772); sh_degree == 0 -> sigmoid of the DC term (:776-777)
        deg = min(self.step // cfg.sh_degree_interval, cfg.sh_degree) if cfg.sh_degree > 0 else -1
        xys = torch.empty((N, 2), device=dev)
        depths = torch.empty((N,), device=dev)
        radii = torch.empty((N,), dtype=torch.int32, device=dev)
        conics = torch.empty((N, 3), device=dev)
        comp = torch.empty((N,), device=dev)
        hit = torch.empty((N,), dtype=torch.int32, device=dev)
        box = torch.empty((N, 4), dtype=torch.int32, device=dev)
        cap = max(self._cap, 1 << 16)
        f32, i32 = torch.float32, torch.int32
        opac = gp["opacities"].reshape(-1)
        total = C.c_int64(0)
        for attempt in range(2):
            ws = self._workspace(N, cap, tiles)
            wsp = C.c_void_p(ws.data_ptr())
            _lib.check(lib.tn_splat_project(C.byref(cam), _ptr(gp["means"], f32, "means"), _ptr(gp["scales"], f32, "scales"), _ptr(gp["quats"], f32, "quats"),
                                            _ptr(opac, f32, "opacities"), _ptr(gp["features_dc"], f32, "features_dc"),
                                            _ptr(gp["features_rest"], f32, "features_rest") if K else None,
                                            _ptr(gp["features_dc_thermal"], f32, "features_dc_thermal"),
                                            _ptr(gp["features_rest_thermal"], f32, "features_rest_thermal") if K else None, N, K, deg, aa,
                                            _ptr(xys, f32, "xys"), _ptr(depths, f32, "depths"), _ptr(radii, i32, "radii"), _ptr(conics, f32, "conics"),
                                            _ptr(comp, f32, "compensation"), _ptr(hit, i32, "num_tiles_hit"), _ptr(box, i32, "tile_box"), wsp, cap, _stream()),
                       "tn_splat_project")
            rc = lib.tn_splat_bin(C.byref(cam), _ptr(depths, f32, "depths"), N, wsp, cap, C.byref(total), _stream())
            if rc == 0:
                break
            if attempt == 0 and total.value > cap:  # the workspace was sized for fewer (Gaussian, tile) pairs: grow once and redo the frame
                cap = int(total.value * 1.25) + 1024
                continue
            _lib.check(rc, "tn_splat_bin")
        self.last_projection = {"xys": xys, "depths": depths, "radii": radii, "conics": conics, "compensation": comp, "num_tiles_hit": hit, "tile_box": box}
        self.last_num_intersections = int(total.value)
        background = bg.to(dev)
        if total.value == 0:  # nothing on screen (splatfacto.py:759-764)
            return {"rgb": background.repeat(H, W, 1), "thermal": torch.full((H, W, 1), cfg.background_thermal, device=dev),
                    "depth": torch.full((H, W, 1), 10.0, device=dev), "accumulation": torch.zeros((H, W, 1), device=dev), "background": background}
        rgbt = torch.empty((H, W, 4), device=dev)
        depth = torch.empty((H, W, 1), device=dev)
        alpha = torch.empty((H, W, 1), device=dev)
        bg4 = (C.c_float * 4)(float(bg[0]), float(bg[1]), float(bg[2]), float(cfg.background_thermal))
        _lib.check(lib.tn_splat_raster(C.byref(cam), N, C.c_void_p(self._ws.data_ptr()), cap, bg4, aa, _ptr(rgbt, f32, "rgbt"), _ptr(depth, f32, "depth"),
                                       _ptr(alpha, f32, "alpha"), _stream()), "tn_splat_raster")
        return {"rgb": rgbt[..., :3], "thermal": rgbt[..., 3:], "depth": depth, "accumulation": alpha, "background": background}
