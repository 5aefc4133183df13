"""Readers / writers for the reference's run logs, so that results can be compared and plotted with the reference's own
tools:

* `progress.csv`  — baselines `CSVOutputFormat` (src/logger.py:101-135): one header line of keys, one row per
  `dump_tabular()`; keys that appear later extend the header and earlier rows are padded with empty cells.
* `*.monitor.csv` — baselines `Monitor` / `ResultsWriter` (src/bench/monitor.py:98-121): a `# {json}` header line with
  `t_start` / `env_id`, then CSV columns r (episode return), l (length), t (seconds since t_start).
Host-side plumbing; no device code.
"""
import csv
import json
import time


class ProgressCsv:
    def __init__(self, filename):
        self.file = open(filename, "w+t")
        self.keys = []

    def writekvs(self, kvs):
        extra = [k for k in kvs.keys() if k not in self.keys]
        if extra:
            self.keys.extend(extra)
            self.file.seek(0)
            lines = self.file.readlines()
            self.file.seek(0)
            self.file.write(",".join(self.keys) + "\n")
            for line in lines[1:]:
                self.file.write(line[:-1] + "," * len(extra) + "\n")
        self.file.write(",".join("" if kvs.get(k) is None else str(kvs.get(k)) for k in self.keys) + "\n")
        self.file.flush()

    def close(self):
        self.file.close()


def read_progress_csv(filename):
    """-> {key: [float or None per row]}"""
    with open(filename) as f:
        rows = list(csv.reader(f))
    keys = rows[0]
    out = {k: [] for k in keys}
    for r in rows[1:]:
        for i, k in enumerate(keys):
            c = r[i] if i < len(r) else ""
            out[k].append(float(c) if c != "" else None)
    return out


class MonitorWriter:
    EXT = "monitor.csv"

    def __init__(self, filename, env_id=None, t_start=None):
        if not filename.endswith(self.EXT):
            filename = filename + "." + self.EXT
        self.t_start = time.time() if t_start is None else t_start
        self.f = open(filename, "wt")
        self.f.write("# %s \n" % json.dumps({"t_start": self.t_start, "env_id": env_id}))
        self.w = csv.DictWriter(self.f, fieldnames=("r", "l", "t"))
        self.w.writeheader()
        self.f.flush()

    def write_episodes(self, ep_rets, ep_lens, t=None):
        """Rows for a batch of finished episodes (e.g. seg["ep_rets"], seg["ep_lens"] of the device generator)."""
        t = round((time.time() if t is None else t) - self.t_start, 6)
        for r, l in zip(ep_rets, ep_lens):
            self.w.writerow({"r": round(float(r), 6), "l": int(l), "t": t})
        self.f.flush()

    def close(self):
        self.f.close()


def read_monitor_csv(filename):
    """-> (header dict, r [E], l [E], t [E]) as lists."""
    with open(filename) as f:
        first = f.readline()
        assert first.startswith("#"), "not a monitor file"
        header = json.loads(first[1:])
        rows = list(csv.DictReader(f))
    return header, [float(x["r"]) for x in rows], [int(x["l"]) for x in rows], [float(x["t"]) for x in rows]
