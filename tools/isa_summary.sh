#!/bin/bash
# usage: tools/isa_summary.sh <asm from -save-temps> <mangled fragment>: resource usage + opcode classes of one kernel
awk -v frag="$2" '
$0 ~ "^_ZN[A-Za-z0-9_]*" frag "[A-Za-z0-9_]*:" {inside=1; next}
inside && /\.end_amdhsa_kernel|^\.Lfunc_end/ {inside=0}
inside && /^\t[a-z]/ {n[$1]++; tot++; if ($1 ~ /^v_/) v++; else if ($1 ~ /^s_/) s++; }
END {printf "total %d valu %d salu %d  s_mov_b32 %d s_load %d  readlane %d writelane %d waitcnt %d\n", tot, v, s, n["s_mov_b32"], n["s_load_dwordx2"]+n["s_load_dwordx4"]+n["s_load_dwordx8"]+n["s_load_dwordx16"]+n["s_load_dword"], n["v_readlane_b32"], n["v_writelane_b32"], n["s_waitcnt"]}' "$1"
grep -A40 "^\s*\.amdhsa_kernel _ZN.*$2" "$1" | grep -E "next_free_vgpr|next_free_sgpr|group_segment|private_segment_fixed" | tr '\n' ' '; echo
grep -B2 -A30 "\.name:.*$2" "$1" | grep -E "sgpr_spill|vgpr_spill|vgpr_count|sgpr_count" | tr '\n' ' '; echo
