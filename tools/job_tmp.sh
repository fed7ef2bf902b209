#!/bin/bash
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force 2>&1 | grep -i " error" 
python - <<PY
import ctypes, torch
torch.zeros(1, device="cuda")
lib = ctypes.CDLL("dagl_amd/csrc/libdagl_ce.so")
print("project16 blocks per CU by the runtime:", lib.dagl_debug_p16_occupancy())
p = torch.cuda.get_device_properties(0)
print(p.name, "CUs", p.multi_processor_count, "shared/block", getattr(p, "shared_memory_per_block", None), "shared/CU", getattr(p, "shared_memory_per_multiprocessor", None), "regs/CU", getattr(p, "regs_per_multiprocessor", None), "max threads/CU", p.max_threads_per_multi_processor)
PY
