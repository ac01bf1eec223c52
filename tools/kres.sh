# register / scratch / code size of every kernel of one .hip file: bash tools/kres.sh csrc/file.hip [grep pattern]
F=$1; P=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c $F -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
cur = {}
for l in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)', l)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        if cur: print(cur)
        cur = {'name': v}
    else: cur[k.split(' ')[0]] = v
if cur: print(cur)
" | grep -E "$P"
