cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/calib
for m in 0 1 2; do
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib/f$m -- scripts/ubench/fetch_calib.exe $m > gpurun_out/calib/f$m.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/calib/r$m -- scripts/ubench/fetch_calib.exe $m > gpurun_out/calib/r$m.log 2>&1
  grep "^mode" gpurun_out/calib/f$m.log
done
python3 - <<'PY'
import csv, glob
for m in (0,1,2):
    for tag in ("f","r"):
        for f in glob.glob("gpurun_out/calib/%s%d/**/*counter_collection.csv" % (tag,m), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_gather" in row["Kernel_Name"]:
                    print("mode", m, row["Counter_Name"], row["Counter_Value"])
PY
