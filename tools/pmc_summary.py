"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: sum of each counter over the kernel's dispatches,
dispatch count.  A counter collected in more than one pass (directory) is averaged over the passes, not added up
(round 1 listed SQ_INSTS_VALU in two passes and read twice the true count)."""
import csv, glob, sys, collections, json, os, re
root = sys.argv[1]
per_pass = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> counter -> pass -> sum
disp = collections.defaultdict(set)
for f in glob.glob(root + "/pmc_*/**/*_counter_collection.csv", recursive=True):
    pass_dir = os.path.relpath(f, root).split(os.sep)[0]
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[\w, ]+>)?)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"].split("(")[0]
        per_pass[k][row["Counter_Name"]][pass_dir] += float(row["Counter_Value"])
        disp[(k, row["Counter_Name"], pass_dir)].add(row["Dispatch_Id"])
res = {}
for k, cs in per_pass.items():
    res[k] = {}
    for c, passes in cs.items():
        n = len(passes)
        res[k][c] = {"sum": sum(passes.values()) / n, "dispatches": max(len(disp[(k, c, p)]) for p in passes), "passes": n}
print(json.dumps(res, indent=1))
