#!/usr/bin/env python3
"""Extracts the log lines the reference's own run printed (the embedded cell output of main_svae.ipynb: the
'| epoch ... | ms/batch ... | loss ...' running lines, the '| end of epoch ... (TRAIN)/(VAL)' blocks of all 25 epochs
and the '| End of training ... (TEST)' block, nb:882-1175) and the numbers in them, and writes them to
tests/golden/ref_log_lines.json.  The fixture is OUTPUT DATA of the reference (what its formatter printed for which
values), not source: tests/test_host_logic.py feeds the parsed values through notebook_api's formatters and compares
bytes.  TEST INFRASTRUCTURE; runs in the build container only (the notebook does not travel to the GPU box).

    python oracle/make_golden_logs.py [/root/reference/main_svae.ipynb]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stdout_lines(nb_path):
    nb = json.load(open(nb_path))
    out = []
    for cell in nb["cells"]:
        for o in cell.get("outputs", []):
            if o.get("output_type") == "stream" and o.get("name") == "stdout":
                out.extend("".join(o["text"]).split("\n"))
    return out


def num(tok):
    """The number a log token stands for: int when it prints as one, else float (str() of the result is the token)."""
    return int(tok) if re.fullmatch(r"-?\d+", tok) else float(tok)


def parse_metrics(rest):
    """' | loss = 409.2443 | NDCG@10 = 17.6805 ... (TRAIN)' -> ([[name, value], ...], tag)"""
    m = re.fullmatch(r"((?: \| [^ |]+ = [^ |]+)+) \((\w+)\)", rest)
    pairs = [p.split(" = ") for p in m.group(1).split(" | ")[1:]]
    return [[k, float(v)] for k, v in pairs], m.group(2)


def main():
    nb_path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/main_svae.ipynb"
    lines = stdout_lines(nb_path)
    batch_lines, epoch_blocks, end_block = [], [], None
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.fullmatch(r"\| epoch +(\d+) \| +(\d+)/ *(\d+) batches \| ms/batch +([\d.]+) \| loss ([\d.]+)", ln)
        if m:
            batch_lines.append(dict(line=ln, epoch=int(m.group(1)), batch=int(m.group(2)), batch_limit=int(m.group(3)),
                                    ms_per_batch=float(m.group(4)), loss=float(m.group(5))))
        elif ln.startswith("| end of epoch") and lines[i + 2].startswith("| end of epoch"):
            block = lines[i - 1:i + 4]                      # rule, TRAIN, rule, VAL, rule
            heads = [re.fullmatch(r"\| end of epoch +(\d+) \| time: +([\d.]+)s(.*)", block[j]) for j in (1, 3)]
            tr, tag_tr = parse_metrics(heads[0].group(3))
            va, tag_va = parse_metrics(heads[1].group(3))
            assert (tag_tr, tag_va) == ("TRAIN", "VAL") and heads[0].group(1) == heads[1].group(1)
            epoch_blocks.append(dict(text="\n".join(block), epoch=int(heads[0].group(1)),
                                     seconds=float(heads[0].group(2)), train=tr, val=va))
            i += 3
        elif ln.startswith("| End of training"):
            block = lines[i - 1:i + 2]
            te, tag = parse_metrics(ln[len("| End of training"):])
            assert tag == "TEST"
            end_block = dict(text="\n".join(block), test=te)
        i += 1
    assert len(epoch_blocks) == 25 and end_block is not None and len(batch_lines) == 125
    out = dict(source="embedded stdout of main_svae.ipynb (cell outputs nb:882-1175)", batch_lines=batch_lines,
               epoch_blocks=epoch_blocks, end_block=end_block)
    path = os.path.join(ROOT, "tests", "golden", "ref_log_lines.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(batch_lines), "running lines,", len(epoch_blocks), "epoch blocks")


if __name__ == "__main__":
    main()
