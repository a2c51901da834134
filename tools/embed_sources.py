"""Writes the device headers as C string literals (roki-fd_amd/build/rkfd_device_src.inc) so that the library carries the sources
rkfdBatchSpecialize hands to hipRTC: no csrc/ or include/ directory is needed beside librkfd_amd.so at run time.
usage: embed_sources.py <out.inc> <header> ...     (include name = path below include/ or csrc/)"""
import os
import sys

out, files = sys.argv[1], sys.argv[2:]
with open(out, "w") as fp:
    names = []
    for i, f in enumerate(files):
        name = f.split("/csrc/")[-1] if "/csrc/" in f else os.path.basename(f)
        names.append(name)
        fp.write("static const char rkfd_src_%d[] =\n" % i)
        for line in open(f):
            fp.write('  "' + line.rstrip("\n").replace("\\", "\\\\").replace('"', '\\"') + '\\n"\n')
        fp.write(";\n")
    fp.write("static const char *const rkfd_src_text[] = { " + ", ".join("rkfd_src_%d" % i for i in range(len(files))) + " };\n")
    fp.write("static const char *const rkfd_src_name[] = { " + ", ".join('"%s"' % n for n in names) + " };\n")
    fp.write("static const int rkfd_src_count = %d;\n" % len(files))
