#!/usr/bin/env python3
"""Per-kernel ISA statistics from `hipcc -S --cuda-device-only` output: instruction counts that matter for this
library (scalar/vector loads, 64-bit multiply-adds, LDS ops, scratch) and register / LDS usage.
usage: isa_stats.py file.s [substring-of-mangled-name]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+):\s*; @\S+\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    c = lambda p: len(re.findall(p, body))
    g = lambda k: (re.search(r'\.set ' + re.escape(name) + r'\.' + k + r', (\d+)', s) or [None, '?'])[1]
    lds = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'.*?group_segment_fixed_size (\d+)', s, re.S)
    print(f"{name[:70]}\n   vgpr {g('num_vgpr')} agpr {g('num_agpr')} sgpr {g('numbered_sgpr')} scratch {g('private_seg_size')} "
          f"lds {lds.group(1) if lds else '?'} | s_load {c('s_load_dword')} global_load {c('global_load')} "
          f"mad_u64 {c('v_mad_u64_u32')} ds_read {c('ds_read')} ds_write {c('ds_write')} scratch_ops {c('scratch_')} "
          f"barrier {c('s_barrier')}")
