import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sdqlpy_amd import abi, engine, frontend, tpch
from sdqlpy_amd import tpch_queries as Q
import reference_shapes as shapes, helpers
from sdqlpy_amd.sdql_lib import sdqlpy_init
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
eng.plan_graphs = int(os.environ.get("G", "2"))
db = tpch.generate(1.0, tables=sorted(tpch.columns_for(["q2"])), columns=tpch.columns_for(["q2"]))
plan = frontend.lower_function(shapes.QUERIES["q2"])
order = sys.argv[1] if len(sys.argv) > 1 else "rs"
for ch in order:
    if ch == "r":
        r = engine.execute_plan(eng, plan, [db[t] for t in shapes.TABLES["q2"]])
    elif ch == "s":
        r = helpers.run_query(eng, "q2", db)
    elif ch == "c":
        eng.clear(); continue
    r = r.wait() if hasattr(r, "wait") else r
    rows = sorted(r.rows())
    print(ch, len(rows), rows[0][3:6], [type(x).__name__ for x in rows[0]], flush=True)
print(eng.stats())
