#!/usr/bin/env python3
"""summarize_experiments.py FILE - the `opt_in_experiments` legs of a bench line as tables.  FILE: a driver record (BENCH_rNN.json: the line sits
under "parsed") or a file whose last line is the JSON line bench.py printed."""
import json
import sys


def load(path):
    txt = open(path).read().strip()
    try:
        d = json.loads(txt)
    except json.JSONDecodeError:
        d = json.loads(txt.splitlines()[-1])
    return d.get("parsed", d)


def main():
    line = load(sys.argv[1])
    ex = line.get("opt_in_experiments")
    if not ex:
        raise SystemExit("no opt_in_experiments in this line")
    print("headline RTF %.2f, decode step (stage_ms_per_token.semantic) %.1f us" % (line["value"], 1000.0 * line["stage_ms_per_token"]["semantic"]))
    print("\nsingle-utterance decode arms (separate processes):")
    base = ex.get("default", {})
    print("%-24s %8s %10s %10s %8s  %s" % ("arm", "RTF", "step@300", "step@640", "vs def", "bits"))
    for name, r in ex.items():
        if not isinstance(r, dict) or "rtf" not in r:
            if isinstance(r, dict) and ("error" in r or "skipped" in r):
                print("%-24s %s" % (name, r.get("error") or r.get("skipped")))
            continue
        s3, s6 = r["decode_step_us"]["300"], r["decode_step_us"]["640"]
        rel = s6 / base["decode_step_us"]["640"] if "decode_step_us" in base else float("nan")
        print("%-24s %8.2f %10.2f %10.2f %7.1f%%  %s" % (name, r["rtf"], s3, s6, 100.0 * (rel - 1.0), r.get("bits_equal_to_the_default_arm")))
    jobs = ex.get("lock_step_jobs", {})
    if jobs:
        print("\nlock-step job arms:")
        for name, r in jobs.items():
            if not isinstance(r, dict):
                continue
            if "prompts_per_s" not in r:
                print("%-30s %s" % (name, r.get("error") or r.get("skipped")))
                continue
            print("%-30s %6.2f prompts/s  bits %-5s stages ms %s" % (name, r["prompts_per_s"], r.get("bits_equal_to_the_default_arm", "-"),
                                                                     {k: round(v) for k, v in r["stage_ms"].items()}))
            if "lock_step_us_by_site" in r:
                print("%-30s   lock step us by site: %s" % ("", r["lock_step_us_by_site"]))
    print("\nexperiment legs took %.0f s" % ex.get("wall_s", float("nan")))


if __name__ == "__main__":
    main()
