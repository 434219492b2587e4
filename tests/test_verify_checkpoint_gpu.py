"""tests/verify_checkpoint.py -- the one-command parity table for a user's real checkpoint (reference README.md:125-139;
strict loading instead of notebooks/infer.py:124-131's strict=False) -- run on the committed tiny checkpoint, from files on
disk exactly as a user would point it at `logs/<experiment>/`."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _project(tmp_path, golden_dir):
    from emoportraits_amd import config
    tiny = torch.load(os.path.join(golden_dir, "tiny_hotpath.pt"), weights_only=False)
    exp = tmp_path / "logs" / "exp"
    (exp / "checkpoints").mkdir(parents=True)
    cfg = config.hot_path_config(overrides=tiny["cfg"])
    with open(exp / "args.txt", "wt") as f:
        for k, v in cfg.items():
            f.write(f"{k}: {v}\n")
    torch.save(tiny["state_dict"], exp / "checkpoints" / "model.pth")
    return exp


def test_verify_checkpoint_on_the_tiny_checkpoint(tmp_path, golden_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import verify_checkpoint
    exp = _project(tmp_path, golden_dir)
    lines = []
    table = verify_checkpoint.verify(exp / "args.txt", exp / "checkpoints" / "model.pth", batch=5, frames=(0, 4), seed=3,
                                     log=lines.append)
    print("\n".join("PARITY verify_checkpoint: " + l for l in lines))
    assert table["ok"] and [r["mode"] for r in table["rows"]] == ["default", "f32", "f16x2"]
    for r in table["rows"]:
        assert set(r["source"]) == {"latents", "source_volume", "pre_canonical", "canonical"} and r["driver"]["u8_same"] > 0.99


def test_verify_checkpoint_command_line_and_strict_loading(tmp_path, golden_dir):
    exp = _project(tmp_path, golden_dir)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "verify_checkpoint.py"), str(exp / "args.txt"),
           str(exp / "checkpoints" / "model.pth"), "--batch", "2", "--frames", "1", "--modes", "default", "--no-source",
           "--json", str(tmp_path / "t.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "inside the stated bounds" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(tmp_path / "t.json")
    sd = torch.load(exp / "checkpoints" / "model.pth")
    sd.pop(next(k for k in sd if k.startswith("decoder_nw")))
    torch.save(sd, exp / "checkpoints" / "broken.pth")
    cmd[3] = str(exp / "checkpoints" / "broken.pth")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "missing" in (r.stdout + r.stderr)
