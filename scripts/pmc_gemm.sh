cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ITERS=3 timeout 300 rocprofv3 --pmc $c --kernel-trace -d /root/repo/gpurun_out/pmc_$c -- python /root/repo/scripts/gemm_one.py > /root/repo/gpurun_out/pmc_$c.log 2>&1
done
cd /root/repo
python - <<'PY'
import glob, sqlite3
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(f"gpurun_out/pmc_{c}/*/*_results.db"):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
        pm = [t for t in tabs if "pmc" in t.lower()]
        print(c, pm[:8])
        for t in pm:
            try:
                cols = [r[1] for r in con.execute(f"pragma table_info({t})")]
                print(" ", t, cols[:14])
            except Exception as e:
                print(" ", t, e)
        try:
            rows = list(con.execute("select k.name, c.name, avg(c.value), count(*) from counters_collection c join kernels k on c.dispatch_id = k.dispatch_id group by 1,2"))
            for r in rows: print("  ", r[0][:60], r[1], r[2], r[3])
        except Exception as e:
            print("  query failed:", e)
PY
