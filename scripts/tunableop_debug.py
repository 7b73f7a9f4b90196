import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import tuning
t = torch.cuda.tunable
t.enable(True); t.tuning_enable(False)
print("file exists", os.path.exists(tuning.RESULTS), tuning.RESULTS)
print("read_file ->", t.read_file(tuning.RESULTS))
print("n results", len(t.get_results()))
print("validators", t.get_validators())
print("is_enabled", t.is_enabled(), "tuning", t.tuning_is_enabled(), "filename", t.get_filename())
x = torch.randn(49250, 768, device="cuda"); w = torch.randn(2304, 768, device="cuda"); b = torch.randn(2304, device="cuda")
for _ in range(3): y = torch.nn.functional.linear(x, w, b)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): y = torch.nn.functional.linear(x, w, b)
e.record(); torch.cuda.synchronize()
print("qkv linear ms", s.elapsed_time(e) / 10)
