# Dev tool: VGPR / scratch use of every kernel of one source file.  usage: bash tools/regs.sh patchconv [-DUDAPOSE_ELEM_F16]
F=${1:-igemm}; shift
mkdir -p /tmp/regs && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}/uda_poseestimation_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c $F.hip -o /tmp/regs/$F.o -save-temps=obj 2>&1 | grep -v "^$" | head -20
python3 - "$F" <<'PY'
import re, glob, sys
f = [x for x in glob.glob('/tmp/regs/%s-hip-amdgcn*.s' % sys.argv[1])][0]
txt = open(f).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    body = m.group(2)
    v = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)
    sp = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)
    print(m.group(1)[:110], 'vgpr', v, 'scratch', sp)
PY
