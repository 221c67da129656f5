"""The CPU oracle over many streams at once (test infrastructure): one process per usable CPU, a stream per job.  The oracle
takes 0.08 s for 256 frames of one stream, so a whole 1024 x 256 batch is some 6 s on 16 cores and a 4096 x 256 one half a
minute: whole batches are compared, not samples."""
import multiprocessing as mp
import os


def _job(args):
    kw, pcm, nfr = args
    from oracle import oracle as O
    enc = O.OracleEncoder(O.default_control(**kw))
    return b"".join(enc.encode_s16(pcm[f * 1152:(f + 1) * 1152]) for f in range(nfr))


def usable_cpus():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return max(1, os.cpu_count() or 1)


def oracle_bytes_many(kws, pcm, nfr, ids=None, block=512):
    """-> {stream id: oracle bitstream} for streams `ids` (default all) of pcm [S, samples, 2]; kws: one control dict or a list"""
    ids = list(range(len(pcm))) if ids is None else [int(i) for i in ids]
    out = {}
    with mp.get_context("spawn").Pool(min(len(ids), usable_cpus(), 32)) as pool:
        for c0 in range(0, len(ids), block):       # in blocks: the jobs' PCM is pickled to the workers
            blk = ids[c0:c0 + block]
            res = pool.map(_job, [((kws[i] if isinstance(kws, list) else kws), pcm[i], nfr) for i in blk], chunksize=4)
            out.update(zip(blk, res))
    return out
