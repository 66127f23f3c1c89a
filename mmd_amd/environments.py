"""2-D maps of the reference (geometry only) and their precomputed SDF grids (host side, init time).

Geometry: deps/torch_robotics/torch_robotics/environments/env_{empty,empty_nowait,highways,conveyor,drop_region}_2d.py.
SDF: MultiRoundedBoxField (primitives.py:312-333) composed with the empty MultiSphereField's constant 1
(primitives.py:109-110) through ObjectField's min (:567-570); grid = GridMapSDF.precompute_sdf
(grid_map_sdf.py:34-63): value and autograd gradient on linspace(lo,hi,ceil(dim/cell))^2.
The reference rebuilds this grid N+1 times per trial (once per planner); here it is built once per map and shared.
"""
import functools

import numpy as np
import torch

MAP_BOXES = {
    "EnvEmpty2D": ([], []),
    "EnvEmptyNoWait2D": ([], []),
    "EnvHighways2D": ([[0, 0.0], [0., 0.875], [0., -0.875], [0.875, 0.0], [-0.875, 0.0], [0.875, 0.875],
                       [0.875, -0.875], [-0.875, 0.875], [-0.875, -0.875]],
                      [[0.5, 0.5], [0.5, 0.25], [0.5, 0.25], [0.25, 0.5], [0.25, 0.5], [0.25, 0.25], [0.25, 0.25],
                       [0.25, 0.25], [0.25, 0.25]]),
    "EnvConveyor2D": ([[0, 0], [0, 0.35], [0, -0.35]], [[0.8, 0.1], [1.0, 0.1], [1.0, 0.1]]),
    "EnvDropRegion2D": ([[0.4, 0.4], [-0.4, 0.4], [0.4, -0.4], [-0.4, -0.4]], [[0.4, 0.4]] * 4),
}
LIMITS = ((-1.0, -1.0), (1.0, 1.0))
SDF_CELL_SIZE = 0.005


def map_sdf(points, map_name):
    x = torch.as_tensor(points, dtype=torch.float32)
    centers, sizes = MAP_BOXES[map_name.replace("ExtraObjects", "")]
    ones = torch.ones_like(x[..., 0])
    if len(centers) == 0:
        return ones
    centers = torch.tensor(centers, dtype=torch.float32)
    sizes = torch.tensor(sizes, dtype=torch.float32)
    radius = torch.min(sizes, dim=-1)[0] * 0.15
    q = torch.abs(x.unsqueeze(-2) - centers.unsqueeze(0)) - (sizes / 2).unsqueeze(0) + radius.unsqueeze(0).unsqueeze(-1)
    max_q = torch.amax(q, dim=-1)
    sdfs = torch.minimum(max_q, torch.zeros_like(max_q)) + torch.linalg.norm(torch.relu(q), dim=-1) - radius.unsqueeze(0)
    return torch.minimum(ones, torch.min(sdfs, dim=-1)[0])


@functools.lru_cache(maxsize=None)
def sdf_grid_texture(map_name, cell_size=SDF_CELL_SIZE):
    """float32 numpy [nx, ny, 4] = (sdf, dsdf/dx, dsdf/dy, 0): the texture layout mmd_guide_desc expects."""
    lo, hi = torch.tensor(LIMITS[0]), torch.tensor(LIMITS[1])
    cmap = torch.ceil(torch.abs(hi - lo) / cell_size).long()
    pts = torch.stack(torch.meshgrid(torch.linspace(lo[0], hi[0], int(cmap[0])),
                                     torch.linspace(lo[1], hi[1], int(cmap[1])), indexing="ij"), dim=-1)
    pts.requires_grad_(True)
    sdf = map_sdf(pts, map_name)
    grad = torch.autograd.grad(sdf.sum(), pts)[0] if sdf.requires_grad else torch.zeros_like(pts)
    tex = torch.zeros(int(cmap[0]), int(cmap[1]), 4)
    tex[..., 0] = sdf.detach()
    tex[..., 1:3] = grad
    return np.ascontiguousarray(tex.numpy())
