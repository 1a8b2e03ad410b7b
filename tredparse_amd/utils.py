"""Host-side helpers mirroring tredparse/utils.py (InputParams :24-51, mkdir :81-104, listify :107-108)."""
import logging
import os
import os.path as op
import shutil


class InputParams:
    """All inputs of one sample x locus unit (tredparse/utils.py:24-51)."""
    KWARGS_LOG = 'log'

    def __init__(self, bam, READLEN, repo, tredName, gender="Unknown", depth=30,
                 clip=False, alts=True, repeatpairs=False, **kwargs):
        self.bam = bam
        self.READLEN = READLEN
        self.tredName = tredName
        self.gender = gender
        self.depth = depth
        self.tred = repo.get(tredName)
        self.clip = clip                # Use clipped reads?
        self.alts = alts                # More exhaustive search?
        self.repeatpairs = repeatpairs  # Include pairs of REPT reads?
        self.kwargs = kwargs
        self.ref = repo.ref

    def getLogLevel(self, defaultLevel='INFO'):
        levelName = self.kwargs.get(InputParams.KWARGS_LOG, defaultLevel)
        return getattr(logging, str(levelName).upper(), defaultLevel)


def mkdir(dirname, overwrite=False, logger=None):
    if op.isdir(dirname):
        if overwrite:
            shutil.rmtree(dirname)
            os.mkdir(dirname)
        else:
            return False
    else:
        try:
            os.mkdir(dirname)
        except OSError:
            os.makedirs(dirname)
    return True


def listify(a):
    return a if isinstance(a, (list, tuple)) else [a]
