"""The parity gates live in oracle/gates.py (one statement, shared with bench.py's parity leg); re-exported here."""
from oracle.gates import (CELL_TOL, DB_GATE, FLOOR_DB, MARGIN_K, NOTCH_ABS, cfar1d_margins, db_map_gate,  # noqa: F401
                          detection_gate, map_cell_gate, margin_eps, mean_level, notch_mask)
