"""Generates tools/micro/leaf_phases_kernel.hip: a copy of mogptk_amd/csrc/leaf.hip with s_memtime markers at the phase boundaries of
k_leaf128 (tile load | per 16-column step: micro-panel factorisation, MFMA trailing update | log-det | diagonal 16x16 inverses |
inverse rows | W store).  Then:
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../mogptk_amd/csrc -I../../include leaf_phases.hip -o leaf_phases && ./leaf_phases"""
import os
here = os.path.dirname(os.path.abspath(__file__))
s = open(os.path.join(here, "..", "..", "mogptk_amd", "csrc", "leaf.hip")).read()
edits = [
    ('namespace mogp {\n\ntypedef double d4_t',
     'namespace mogp {\n__device__ unsigned long long g_leaf_t[64];\n#define MARK(i) do { __syncthreads(); if (threadIdx.x == 0) g_leaf_t[i] = __builtin_readcyclecounter(); } while (0)\n\ntypedef double d4_t'),
    ('    __builtin_amdgcn_s_setprio(3);        // serial critical path: outrank co-resident trailing-update waves\n', '    __builtin_amdgcn_s_setprio(3);\n    MARK(0);\n'),
    ('    int fail = -1;\n    for (int sb = 0; sb < 8; ++sb) {', '    MARK(1);\n    int fail = -1;\n    for (int sb = 0; sb < 8; ++sb) {'),
    ('        // ---- P3, the part done right away:', '        MARK(2 + 2 * sb);\n        // ---- P3, the part done right away:'),
    ('                    if ((cnt & 3) == wave) lf_update(M, i, j, sb, lane);\n        }\n        __syncthreads();\n    }\n',
     '                    if ((cnt & 3) == wave) lf_update(M, i, j, sb, lane);\n        }\n        __syncthreads();\n        MARK(3 + 2 * sb);\n    }\n'),
    ('    // ---- TRTRI: diagonal 16x16 inverses, one column per lane', '    MARK(20);\n    // ---- TRTRI: diagonal 16x16 inverses, one column per lane'),
    ('    // ---- TRTRI, off-diagonal blocks:', '    MARK(21);\n    // ---- TRTRI, off-diagonal blocks:'),
    ('    // diagonal 16 x 16 blocks of the tile inverse (from LDS)', '    MARK(22);\n    // diagonal 16 x 16 blocks of the tile inverse (from LDS)'),
    ('        *reinterpret_cast<d2_t*>(Wt + r * MOGP_TILE + c) = v;\n    }\n}\n', '        *reinterpret_cast<d2_t*>(Wt + r * MOGP_TILE + c) = v;\n    }\n    MARK(23);\n}\n'),
]
for old, new in edits:
    assert old in s, "leaf.hip changed: update the marker anchors (%r)" % old[:50]
    s = s.replace(old, new, 1)
open(os.path.join(here, "leaf_phases_kernel.hip"), "w").write(s)
print("wrote leaf_phases_kernel.hip with %d markers" % s.count("MARK("))
