"""Per-kernel register / LDS / scratch table of the library, from hipcc's kernel-resource-usage remarks
(no GPU needed):  python tools/kernel_resources.py > profiles/rN_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = os.path.join(tempfile.gettempdir(), "bpe_api_res.o")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-fPIC",
       f"-I{ROOT}/include", f"-I{ROOT}/minbpe_amd/csrc", f"{ROOT}/minbpe_amd/csrc/bpe_api.hip", "-o", obj,
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch_B_per_lane", r"ScratchSize \[bytes/lane\]"),
        ("waves_per_simd", r"Occupancy \[waves/SIMD\]"), ("lds_B", r"LDS Size \[bytes/block\]")]
print("kernel," + ",".join(k for k, _ in keys))
for b, n in zip(blocks, dem):
    vals = []
    for _, pat in keys:
        m = re.search(pat + r": (\d+)", b)
        vals.append(m.group(1) if m else "")
    n = re.sub(r"\(.*", "", n)
    print(f'"{n}",' + ",".join(vals))
