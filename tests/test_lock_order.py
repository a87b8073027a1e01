"""Static check of the host library's lock order: wherever one function takes both the
device's staging pool (StageScope -> stage_mu) and its enqueue mutex (enq_mu), the staging pool comes first.  The
deadlock this guards against needs a GPU and two threads to show; the order itself is visible in the text."""
import glob
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc")


def _functions(text):
    """(start, end) spans of top-level brace blocks that follow a ')' -- function bodies, macro bodies included"""
    spans, depth, start = [], 0, None
    for i, ch in enumerate(text):
        if ch == "{":
            if depth == 0:
                start = i
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0 and start is not None:
                spans.append((start, i))
                start = None
    return spans


def test_staging_pool_is_locked_before_the_enqueue_mutex():
    seen = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.inc"))):
        if os.path.basename(path).startswith("tower_vm_"):
            continue  # generated tables
        text = re.sub(r"//[^\n]*", "", open(path).read())
        # namespaces / extern "C" blocks are brace blocks too: look inside them, one level at a time
        work = [text]
        while work:
            t = work.pop()
            for a, b in _functions(t):
                body = t[a + 1:b]
                head = t[max(0, a - 200):a]
                if re.search(r"namespace\s+\w*\s*$|extern\s+\"C\"\s*$", head.rstrip()):
                    work.append(body)
                    continue
                i_stage = body.find("StageScope ")
                i_enq = body.find("enq_mu")
                if i_stage >= 0 and i_enq >= 0:
                    seen += 1
                    assert i_stage < i_enq, f"{os.path.basename(path)}: enqueue mutex taken before the staging pool"
    assert seen >= 2  # the MSM and poly_eval host paths at least
