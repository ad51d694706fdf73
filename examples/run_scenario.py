#!/usr/bin/env python3
"""CLI-shaped caller of the drop-in simulator: the reference's run flow (backend/server.py:231-266 and the
SSE body :341-394; backend/main.py:54-133 follows the same sequence) through
`monte_carlo_retirement_amd.results.run_scenario` — load a scenario JSON, search the minimum working months,
switch to the final seed stream, run the final batch, print the response document.

    python examples/run_scenario.py scenarios/jorge.json --seed 12345 --rng numpy
    python examples/run_scenario.py scenarios/config.json --paths 10000000 --working-months 233 --compact
    python examples/run_scenario.py scenarios/config.json --events --full > response.json

`--rng numpy` uses the reference's own NumPy stream (same seed -> the reference's numbers); `--rng philox`
(default) the engine's counter-based stream.  `--compact` assembles the document from device-side aggregates
only (no per-path lists: for batches far beyond the UI's).  `--events` writes the progress events the SSE
endpoint would stream to stderr, one JSON per line.  Without `--full` only the `summary` block (plus timings
and sizes) is printed."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)

from monte_carlo_retirement_amd import Config, load_config_from_json  # noqa: E402
from monte_carlo_retirement_amd import results as R  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--rng", choices=["philox", "numpy"], default="philox")
    ap.add_argument("--paths", type=int, default=None, help="override num_simulations_main")
    ap.add_argument("--search-paths", type=int, default=None, help="override num_simulations_search")
    ap.add_argument("--working-months", type=int, default=None, help="skip the search")
    ap.add_argument("--compact", action="store_true", help="device-aggregated document (large batches)")
    ap.add_argument("--events", action="store_true", help="progress events to stderr")
    ap.add_argument("--full", action="store_true", help="print the whole response document")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if world > 1:  # one process per GPU (python -m torch.distributed.run --nproc-per-node N examples/run_scenario.py ...)
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        dist.init_process_group(os.environ.get("MCR_BACKEND", "nccl"))
        if args.compact:
            ap.error("--compact is a single-process document")
    raw = load_config_from_json(args.config)
    if args.paths:
        raw["num_simulations_main"] = args.paths
    if args.search_paths:
        raw["num_simulations_search"] = args.search_paths
    config = Config(**raw)

    marks = {"t0": time.perf_counter()}
    seen = []

    def emit(event: dict) -> None:
        seen.append(event["type"])
        if event["type"] in ("search_complete", "error") or (event["type"] == "phase" and event["phase"] == "final_sim"):
            marks.setdefault("search_done", time.perf_counter())
        if args.events and rank0 and event["type"] != "result":
            print(json.dumps(event), file=sys.stderr)

    builder = R.compact_result if args.compact else R.build_result
    doc = R.run_scenario(config, args.working_months, emit=emit, result_builder=builder,
                         main_seed_override=args.seed, rng=args.rng)
    t_end = time.perf_counter()
    rc = 0
    if doc is None:
        rc = 1
        out = {"error": "see the last event", "events": seen[-3:]}
    elif args.full:
        out = doc
    else:
        out = {
            "scenario": doc["scenario"], "rng": args.rng, "summary": doc["summary"],
            "search_probes": seen.count("search_iter"),
            "trajectory_points": len(doc["trajectory"]["years"]),
            "sample_paths": len(doc["trajectory"]["sample_paths"]),
            "failure_count": doc["ruin_histogram"]["failure_count"],
            "document_bytes": len(json.dumps(doc)),
            "seconds": {"search": round(marks.get("search_done", t_end) - marks["t0"], 3),
                        "final_run_and_document": round(t_end - marks.get("search_done", t_end), 3)},
        }
    if rank0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
