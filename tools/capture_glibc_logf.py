"""Prints the logf table of the GNU libc this machine runs (sysdeps/ieee754/flt-32/logf_data.c: 16 x {invc, logc},
ln2, three polynomial coefficients) as found in libm.so.6 - the constants hmp3_amd/csrc/hx_libm32.h restates.
python tools/capture_glibc_logf.py [path to libm.so.6]"""
import struct
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
blob = open(path, "rb").read()
ln2 = struct.pack("<d", float.fromhex("0x1.62e42fefa39efp-1"))
at = blob.find(ln2)
while at >= 0:
    tab = struct.unpack("<32d", blob[at - 256:at]) if at >= 256 else ()
    if tab and tab[18] == 1.0 and tab[19] == 0.0:       # the subinterval that contains 1: {1, 0}
        for k in range(16):
            print("%2d  invc 0x%016x  logc 0x%016x   %s %s" % (k, *struct.unpack("<2Q", blob[at - 256 + 16 * k:at - 240 + 16 * k]), tab[2 * k].hex(), tab[2 * k + 1].hex()))
        print("ln2 ", struct.unpack("<d", blob[at:at + 8])[0].hex())
        print("poly", [x.hex() for x in struct.unpack("<3d", blob[at + 8:at + 32])])
        break
    at = blob.find(ln2, at + 1)
else:
    raise SystemExit("logf table not found in " + path)
