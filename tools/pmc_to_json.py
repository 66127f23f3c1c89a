#!/usr/bin/env python
"""gpurun_out/<tag>_pmc*.txt (tools/gpu_pmc.sh) -> profiles/pmc_latest.json, the committed PMC figures bench.py quotes
(`roofline.traffic`, `roofline.mfma_busy_pmc`, `roofline_step_kernel.pmc`).  Usage: pmc_to_json.py <unet pmc txt> <n_traj> [<bench pmc txt> [<unet pmc txt of the whole 2048-trajectory batch>]]"""
import json
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"# (\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"(\S.*?)\s+n=\s*(\d+) mean=\s*([\d.]+)", line)
        if m and cur:
            out.setdefault(m.group(1).strip(), {})[cur] = float(m.group(3))
    return out


unet_txt, n_traj = sys.argv[1], int(sys.argv[2])
u = parse(unet_txt)
k = next(v for kk, v in u.items() if "unet_kernel" in kk)
doc = {"source": f"rocprofv3 --pmc passes (separate runs, --kernel-trace only; tools/gpu_pmc.sh) over tools/unet_forward_loop.py {n_traj}: {unet_txt}",
       "UNET": {"FETCH_SIZE_KiB": k["FETCH_SIZE"], "WRITE_SIZE_KiB": k["WRITE_SIZE"], "trajectories_per_launch": n_traj,
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
                "mfma_busy_frac": k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
                "SQ_INSTS_VALU": k.get("SQ_INSTS_VALU"), "SQ_INSTS_MFMA": k.get("SQ_INSTS_MFMA"),
                "SQ_WAVE_CYCLES": k.get("SQ_WAVE_CYCLES"), "SQ_ACTIVE_INST_ANY": k.get("SQ_ACTIVE_INST_ANY"),
                "SQ_WAIT_ANY": k.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": k.get("SQ_WAIT_INST_ANY"),
                "SQ_LDS_BANK_CONFLICT": k.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": k.get("SQ_LDS_IDX_ACTIVE")}}
if len(sys.argv) > 3:
    b = parse(sys.argv[3])
    for kk, v in b.items():
        if "ddpm_guide_kernel<8" in kk or ("ddpm_guide_kernel" in kk and "GUIDE" not in doc and v.get("SQ_INSTS_VALU", 0) > 1e6):
            doc["GUIDE"] = {"kernel": kk, "source": f"rocprofv3 --pmc passes over a bench.py round (guided step kernel, one launch at a time): {sys.argv[3]}",
                            "FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB": v.get("WRITE_SIZE"),
                            "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_INSTS_LDS": v.get("SQ_INSTS_LDS"),
                            "SQ_WAVE_CYCLES": v.get("SQ_WAVE_CYCLES"), "SQ_ACTIVE_INST_ANY": v.get("SQ_ACTIVE_INST_ANY"),
                            "SQ_WAIT_ANY": v.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": v.get("SQ_WAIT_INST_ANY"),
                            "SQ_ACTIVE_INST_VALU": v.get("SQ_ACTIVE_INST_VALU"),
                            "valu_issue_frac": (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]) if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_WAVE_CYCLES") else None}
if len(sys.argv) > 4:
    w = next(v for kk, v in parse(sys.argv[4]).items() if "unet_kernel" in kk)
    doc["UNET"]["whole_batch_2048"] = {"mfma_busy_frac": w["SQ_VALU_MFMA_BUSY_CYCLES"] / (w["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
                                       "source": f"same passes over tools/unet_forward_loop.py 2048: {sys.argv[4]}"}
json.dump(doc, open("profiles/pmc_latest.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
