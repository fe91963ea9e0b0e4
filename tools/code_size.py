"""Code bytes per kernel family of a library or object file (gfx950 code objects, `llvm-readelf -s`; every kernel is counted with its
descriptor symbol, so counts are 2 x kernels).  Round 6, build time: which families fill the library (profiles/r06_build_time_report.txt).
  python tools/code_size.py [idsp_amd/lib/libidsp_hip.so | some.o]"""
import sys, subprocess, re, collections, tempfile, os, glob, shutil
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp(prefix="idsp_sz_")
try:
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "extract_co.py"), (sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "idsp_amd", "lib", "libidsp_hip.so")), tmp], check=True, capture_output=True)
    fam, cnt = collections.Counter(), collections.Counter()
    for co in glob.glob(tmp + "/co*.o"):
        out = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", co], capture_output=True, text=True).stdout
        for line in out.splitlines():
            p = line.split()
            if len(p) >= 8 and p[3] == "FUNC" and p[7].startswith("_ZN4idsp"):
                mm = re.search(r"\d+(stream_[a-z_]+|cic_[a-z_]+|hbf_[a-z_]+|lockin_[a-z_]+|fm_disc[a-z_]*|[a-z_0-9]+kernel)", p[7])
                k = mm.group(1) if mm else "other"
                fam[k] += int(p[2]); cnt[k] += 1
    for k, v in fam.most_common(30):
        print(f"{k:36s} {cnt[k]:5d} kernels {v/1e6:8.2f} MB  avg {v/cnt[k]/1024:7.1f} KiB")
    print("total MB", sum(fam.values()) / 1e6)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
