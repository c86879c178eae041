"""Grammar of the benchmark logs (the outfile.log / nq*.log contract).

The reference's plotting scripts define the grammar by how they parse
(benchmark05/postprocess.py:9-21, benchmark01/postprocess.py:9-14):
  * keep lines that contain both KEY ("nelmt" or "Size") and UNIT ("DOF/s" or "GB/s");
  * x = float(line.split()[1]); y-columns = [float(t) for t in line.split()[3:]];
  * every kept line must have the same number of columns, at most len(labels)
    (11 for benchmark04/05, 5 for benchmark01-03), else labels[i] raises IndexError;
  * title = last line containing "NQ =".
This module restates that grammar (own code) so both the reference's logs and ours can be parsed,
validated and plotted by the same functions.
"""
from dataclasses import dataclass, field

MAX_COLUMNS = {"DOF/s": 11, "GB/s": 5}


@dataclass
class BenchLog:
    kind: str                      # "DOF/s" (benchmark04/05) or "GB/s" (benchmark01-03)
    title: str = ""
    sizes: list = field(default_factory=list)      # x axis (nelmt or Size)
    values: list = field(default_factory=list)     # values[row][col]
    norms: list = field(default_factory=list)      # norms[row][col]
    case_lines: list = field(default_factory=list)

    @property
    def ncols(self):
        return len(self.values[0]) if self.values else 0

    def column(self, c):
        return [row[c] for row in self.values]


def parse_log(text):
    """Parse a log (str).  Raises ValueError when the text violates the grammar."""
    lines = text.splitlines()
    kind = None
    for ln in lines:
        if "nelmt" in ln and "DOF/s" in ln:
            kind = "DOF/s"
            break
        if "Size" in ln and "GB/s" in ln:
            kind = "GB/s"
            break
    if kind is None:
        raise ValueError("no data lines (neither 'nelmt ... DOF/s' nor 'Size ... GB/s')")
    key = "nelmt" if kind == "DOF/s" else "Size"
    log = BenchLog(kind=kind)
    for ln in lines:
        if "NQ =" in ln:
            log.title = ln.strip()
        if key in ln and kind in ln:
            tok = ln.split()
            log.sizes.append(float(tok[1]))
            log.values.append([float(t) for t in tok[3:]])
        elif key in ln and "norm:" in ln:
            log.norms.append([float(t) for t in ln.split()[3:]])
        elif key in ln and "Case:" in ln:
            log.case_lines.append(ln)
    ncols = {len(v) for v in log.values}
    if len(ncols) != 1:
        raise ValueError(f"data lines have differing column counts: {sorted(ncols)}")
    if log.ncols > MAX_COLUMNS[kind]:
        raise ValueError(f"{log.ncols} columns; the reference's postprocess.py has labels for "
                         f"{MAX_COLUMNS[kind]}")
    if log.norms and any(len(n) != log.ncols for n in log.norms):
        raise ValueError("norm lines and data lines disagree on the column count")
    return log


def parse_file(path):
    with open(path) as fh:
        return parse_log(fh.read())
