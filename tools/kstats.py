"""Prints a rocprofv3 kernel_stats.csv (found under the given directory) as: short kernel name, calls, average us."""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(files[0])))[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-70s calls %6s  avg %9.1f us  total %6.1f%%" % (name[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r.get("Percentage", 0) or 0)))
