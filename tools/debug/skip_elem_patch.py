# TEMPORARY timing experiment, never committed applied: rewrites elem.hip / dense.hip / gemm.hip so that CMOOP_DEBUG_SKIP_ELEM=1 drops every
# non-GEMM launch (=2: also split-K combine and the flip-transpose) -- results are garbage, only the wall time means something.
# Apply, make, run tools/debug/skip_run.sh on the GPU, then `git checkout` the three files and rebuild.
import re
root='/root/repo/cmoop_audio_processing_amd/csrc/'
hdr='''
// ---- TEMPORARY timing experiment (never committed): CMOOP_DEBUG_SKIP_ELEM=1 drops every launch of this file
#include <cstdlib>
static inline int dbg_skip_level() { static const int v = getenv("CMOOP_DEBUG_SKIP_ELEM") ? atoi(getenv("CMOOP_DEBUG_SKIP_ELEM")) : 0; return v; }
#define ELEM_LAUNCH(...) do { if (dbg_skip_level() < LEVEL_OF_FILE) hipLaunchKernelGGL(__VA_ARGS__); } while (0)
'''
for f,level in (('elem.hip',1),('dense.hip',1)):
    s=open(root+f).read()
    s=s.replace('hipLaunchKernelGGL(','ELEM_LAUNCH(')
    # insert header after the last #include
    idx=[m.end() for m in re.finditer(r'^#include.*$', s, re.M)][-1]
    s=s[:idx]+hdr.replace('LEVEL_OF_FILE',str(level))+s[idx:]
    open(root+f,'w').write(s)
# gemm.hip: level 2 skips splitk_combine and flip
s=open(root+'gemm.hip').read()
idx=[m.end() for m in re.finditer(r'^#include.*$', s, re.M)][-1]
s=s[:idx]+hdr.replace('LEVEL_OF_FILE','2')+s[idx:]
s=s.replace('hipLaunchKernelGGL(splitk_combine_kernel','ELEM_LAUNCH(splitk_combine_kernel').replace('hipLaunchKernelGGL(flip_transpose_all_kernel','ELEM_LAUNCH(flip_transpose_all_kernel')
open(root+'gemm.hip','w').write(s)
