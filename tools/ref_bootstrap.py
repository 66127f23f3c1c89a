"""Bootstrap for importing the yoraish/mmd reference (read-only, /root/reference) in THIS container.

Used only by tools/make_golden.py and tools/check_oracle_vs_reference.py to generate / verify golden
vectors.  Nothing under tests/ (gpu marker), bench.py or the product package imports this: the
reference does not exist on the GPU box.  See SURVEY.md Appendix B.
"""
import os
import sys
import types

REF = os.environ.get("MMD_REFERENCE", "/root/reference")


def bootstrap():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    for p in ("", "/deps/torch_robotics", "/deps/motion_planning_baselines", "/deps/experiment_launcher"):
        q = REF + p
        if q not in sys.path:
            sys.path.insert(0, q)
    if "git" not in sys.modules:
        git = types.ModuleType("git")

        class _Repo:  # git.Repo('.', search_parent_directories=True).working_dir
            def __init__(self, *a, **k):
                self.working_dir = "/tmp"

        git.Repo, git.InvalidGitRepositoryError = _Repo, Exception
        sys.modules["git"] = git
    if "wandb" not in sys.modules:
        sys.modules["wandb"] = types.ModuleType("wandb")
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    import matplotlib
    matplotlib.use("Agg")
