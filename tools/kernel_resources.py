"""VGPR / SGPR / scratch / LDS / kernarg bytes of every gfx950 kernel in libapgp.so (llvm-readelf --notes of the code objects
in its fat binary).  Usage: python tools/kernel_resources.py [substring]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_store_hazard import code_objects, LLVM
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "approxposterior_amd", "csrc", "libapgp.so")
want = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as wd:
    for co in code_objects(lib, wd):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in txt.split("  - .agpr_count")[1:]:
            get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
            name = get("name")
            if want in name:
                print("%-70s vgpr %s agpr %s sgpr %s scratch %s lds %s kernarg %s" % (name[:70], get("vgpr_count"), blk.split()[0].strip(": "), get("sgpr_count"),
                      get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("kernarg_segment_size")))
