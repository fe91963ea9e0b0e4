"""Extract the gfx950 code objects of a library or object file (compressed or plain offload bundles) into a directory, for llvm-objdump /
llvm-readelf:  python tools/extract_co.py LIB OUTDIR   (OUTDIR is emptied first)"""
import os, struct, subprocess, sys
LLVM = "/opt/rocm/lib/llvm/bin"
lib, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
for f in os.listdir(out):
    os.remove(os.path.join(out, f))
subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, out + "/fat.bin"], check=True)
data = open(out + "/fat.bin", "rb").read()
pos = data.find(b"CCOB")
n = 0
if pos < 0:
    # uncompressed bundle
    open(out + "/b0.bin", "wb").write(data)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={out}/b0.bin", f"--output={out}/co0.o"], check=True, capture_output=True)
    n = 1
while pos >= 0:
    size = struct.unpack("<Q", data[pos + 8:pos + 16])[0]
    open(f"{out}/b{n}.bin", "wb").write(data[pos:pos + size])
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={out}/b{n}.bin", f"--output={out}/co{n}.o"], check=True, capture_output=True)
    os.remove(f"{out}/b{n}.bin")
    n += 1
    pos = data.find(b"CCOB", pos + size)
os.remove(out + "/fat.bin")
print(n, "code objects in", out)
