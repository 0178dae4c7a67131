#!/bin/bash
# VGPRs / scratch / LDS of the kernels in a built libgcmcore.so: bash tools/tools_kernel_regs.sh [lib.so] [name filter]
LIB=$(readlink -f ${1:-$(dirname $0)/../gcmiipy_amd/lib/libgcmcore.so}); PAT=${2:-pe_}
T=$(mktemp -d); cd $T
B=/opt/rocm/lib/llvm/bin
# the device code: one clang offload bundle per translation unit, concatenated in the .hip_fatbin section
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB fat.bin
python3 - "$PAT" <<'PY'
import sys,re,subprocess
pat=sys.argv[1]
B='/opt/rocm/lib/llvm/bin/'
data=open('fat.bin','rb').read()
magic=b'__CLANG_OFFLOAD_BUNDLE__'
pos=[m.start() for m in re.finditer(magic,data)]+[len(data)]
for n,(a,b) in enumerate(zip(pos,pos[1:])):
    open('b%d.bin'%n,'wb').write(data[a:b])
    subprocess.run([B+'clang-offload-bundler','--unbundle','--type=o','--input=b%d.bin'%n,'--targets=hipv4-amdgcn-amd-amdhsa--gfx950','--output=d%d.co'%n],check=True)
    txt=subprocess.run([B+'llvm-readelf','--notes','d%d.co'%n],capture_output=True,text=True).stdout
    for blk in txt.split('.agpr_count:')[1:]:
        g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
        d=subprocess.run(['c++filt',g('name')],capture_output=True,text=True).stdout.strip()
        if pat in d:
            print('%-100s vgpr %s agpr %s spill %s scratch %s lds %s'%(d[:100],g('vgpr_count'),blk.split()[0],g('vgpr_spill_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
PY
rm -rf $T
