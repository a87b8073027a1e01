"""Static check of the host library's lock order: wherever one function takes both the
device's staging pool (StageScope) and its enqueue mutex (enq_mu), the staging pool comes first.  The deadlock this
guards against needs a GPU and two threads to show; the order itself is visible in the text.  Since round 6 a host-buffer
call holds a POOL (its own stream and buffers), not the device's one staging mutex, and the enqueue mutex only for the
enqueue (inside run() / the _dev entry points): the MSM, poly-eval and scalar-Horner host wrappers, which used to hold
enq_mu across their copies, must no longer name it."""
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
    # round 6: the host wrappers that held the enqueue mutex across upload / run / download no longer take it themselves
    for name, fn in (("msm.cuh", "run_host_single"), ("msm.cuh", "poly_eval_host_single"), ("scalar_poly.hip", "eval_host")):
        text = re.sub(r"//[^\n]*", "", open(os.path.join(CSRC, name)).read())
        hits = [m.start() for m in re.finditer(r"StageScope sc_\(ctx\)", text)]
        assert hits, name
        for h in hits:
            # up to the wrapper's download (the end of the copies): no enqueue mutex in between
            end = text.find("download(", h)
            assert end > h and "enq_mu" not in text[h:end], f"{name}: a host wrapper holds enq_mu across its copies again"
