"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: sum of each counter, dispatch count."""
import csv, glob, sys, collections, json, re
root = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(root + "/pmc_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[\w, ]+>)?)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"].split("(")[0]
        out[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
res = {}
for k, cs in out.items():
    res[k] = {c: {"sum": v, "dispatches": len(disp[(k, c)])} for c, v in cs.items()}
print(json.dumps(res, indent=1))
