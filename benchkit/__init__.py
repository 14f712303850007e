"""Support code of bench.py (the measurement harness; not part of the product package).

    workloads  the BASELINE.json configurations as synthetic jobs: sigma schedules, stub backbone, masks, one schedule pass
    parity     the schedule pass against the CPU oracle that precedes every timed region
    roofline   bytes model of the steady launch, per-dispatch event timing, lookups of the committed rocprofv3 profiles
    cpu        the cpu_baseline leg (the oracle timed on this box's host cores, bounded sample)
    ranks      starting N ranks without a launcher, the per-rank report and its summary
    line       the one JSON line: size bound, side-car file, stdout claim
    extras     secondary measurements (side-car only; `bench.py --extras 1`)
"""
