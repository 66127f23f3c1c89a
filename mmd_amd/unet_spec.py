"""Parameter layout of the reference TemporalUnet, as an ordered (key, shape) list.

Mirrors the module registration order of `TemporalUnet.__init__`
(reference mmd/models/diffusion_models/temporal_unet.py:25-119) and of `ResidualTemporalBlock`,
`Conv1dBlock`, `Downsample1d`, `Upsample1d`, `TimeEncoder` (mmd/models/layers/layers.py:232-358) so that a
released `ema_model_current_state_dict.pth` (keys prefixed `model.`) drops in.
No torch import: used by the weight packer, the synthetic weight generator and the oracle alike.
"""
from collections import OrderedDict

UNET_DIM_MULTS = {0: (1, 2, 4), 1: (1, 2, 4, 8)}   # temporal_unet.py:17-20
TIME_EMB_DIM = 32                                   # temporal_unet.py:31, TimeEncoder(32, time_emb_dim) :71
KERNEL_SIZE = 5                                     # layers.py:328


def group_norm_n_groups(n_channels, target_n_groups=8):
    """layers.py:392-398."""
    if n_channels < target_n_groups:
        return 1
    for n_groups in range(target_n_groups, target_n_groups + 10):
        if n_channels % n_groups == 0:
            return n_groups
    return 1


def level_dims(state_dim=4, unet_input_dim=32, dim_mults=(1, 2, 4)):
    dims = [state_dim] + [unet_input_dim * m for m in dim_mults]
    return list(zip(dims[:-1], dims[1:]))


def _rtb(spec, prefix, cin, cout):
    for b, (ci, co) in enumerate(((cin, cout), (cout, cout))):
        spec[f"{prefix}.blocks.{b}.block.0.weight"] = (co, ci, KERNEL_SIZE)
        spec[f"{prefix}.blocks.{b}.block.0.bias"] = (co,)
        spec[f"{prefix}.blocks.{b}.block.2.weight"] = (co,)
        spec[f"{prefix}.blocks.{b}.block.2.bias"] = (co,)
    spec[f"{prefix}.cond_mlp.1.weight"] = (cout, TIME_EMB_DIM)
    spec[f"{prefix}.cond_mlp.1.bias"] = (cout,)
    if cin != cout:
        spec[f"{prefix}.residual_conv.weight"] = (cout, cin, 1)
        spec[f"{prefix}.residual_conv.bias"] = (cout,)


def unet_param_spec(state_dim=4, unet_input_dim=32, dim_mults=(1, 2, 4)):
    """OrderedDict key -> shape, in the reference's state_dict order (without the `model.` prefix)."""
    in_out = level_dims(state_dim, unet_input_dim, dim_mults)
    n_res = len(in_out)
    spec = OrderedDict()
    spec["time_mlp.encoder.1.weight"] = (TIME_EMB_DIM * 4, TIME_EMB_DIM)
    spec["time_mlp.encoder.1.bias"] = (TIME_EMB_DIM * 4,)
    spec["time_mlp.encoder.3.weight"] = (TIME_EMB_DIM, TIME_EMB_DIM * 4)
    spec["time_mlp.encoder.3.bias"] = (TIME_EMB_DIM,)
    for ind, (din, dout) in enumerate(in_out):
        _rtb(spec, f"downs.{ind}.0", din, dout)
        _rtb(spec, f"downs.{ind}.1", dout, dout)
        if ind < n_res - 1:
            spec[f"downs.{ind}.4.conv.weight"] = (dout, dout, 3)
            spec[f"downs.{ind}.4.conv.bias"] = (dout,)
    for ind, (din, dout) in enumerate(reversed(in_out[1:])):
        _rtb(spec, f"ups.{ind}.0", dout * 2, din)
        _rtb(spec, f"ups.{ind}.1", din, din)
        # is_last = ind >= n_res - 1 is never true for the ups loop (temporal_unet.py:100-101): always Upsample1d
        spec[f"ups.{ind}.4.conv.weight"] = (din, din, 4)     # ConvTranspose1d layout [in, out, k]
        spec[f"ups.{ind}.4.conv.bias"] = (din,)
    mid = in_out[-1][1]
    _rtb(spec, "mid_block1", mid, mid)
    _rtb(spec, "mid_block2", mid, mid)
    spec["final_conv.0.block.0.weight"] = (unet_input_dim, unet_input_dim, KERNEL_SIZE)
    spec["final_conv.0.block.0.bias"] = (unet_input_dim,)
    spec["final_conv.0.block.2.weight"] = (unet_input_dim,)
    spec["final_conv.0.block.2.bias"] = (unet_input_dim,)
    spec["final_conv.1.weight"] = (state_dim, unet_input_dim, 1)
    spec["final_conv.1.bias"] = (state_dim,)
    return spec
