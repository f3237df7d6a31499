"""BwtMergeSort.computeBwt mirror (/root/reference/src/bwtb3m.cpp:63) over b3m_compute_bwt."""
import ctypes
import sys

from . import _lib
from ._lib import B3MError, lib
from .options import BwtMergeSortOptions


class BwtMergeSortResult:
    """file names produced for outputfilename X.bwt (/root/reference/src/checkbwt.cpp:39-53)"""

    def __init__(self, outfn, textfn):
        prefix = outfn[:-4] if outfn.endswith(".bwt") else outfn
        self.textfn = textfn
        self.bwtfn = outfn
        self.histfn = prefix + ".hist"
        self.metafn = prefix + ".meta"
        self.preisafn = prefix + ".preisa"


class BwtMergeSort:
    @staticmethod
    def computeBwt(options, logstr=None):
        """options: BwtMergeSortOptions.  Errors raise B3MError (the reference throws
        LibMausException).  logstr: a file object with fileno() (default stderr)."""
        if not isinstance(options, BwtMergeSortOptions):
            raise TypeError("options must be BwtMergeSortOptions")
        o = _lib.Options()
        lib().b3m_options_init(ctypes.byref(o))
        o.input_fn = options.fn.encode()
        o.inputtype = options.inputtype.encode()
        o.output_fn = options.outfn.encode()
        o.tmpprefix = options.tmpfilenamebase.encode()
        o.sparsetmpprefix = options.sparsetmpfilenamebase.encode()
        o.mem = options.mem
        o.sasamplingrate = options.sasamplingrate
        o.isasamplingrate = options.isasamplingrate
        o.largelcpthres = options.largelcpthres
        o.numthreads = options.numthreads
        o.bwtonly = options.bwtonly
        o.copyinputtomemory = options.copyinputtomemory
        o.verbose = options.verbose
        o.blocksize = options.blocksize
        o.num_gpus = int(getattr(options, "numgpus", 1))
        fd = (logstr or sys.stderr).fileno()
        if lib().b3m_compute_bwt(ctypes.byref(o), fd) != 0:
            raise B3MError(lib().b3m_last_error().decode("utf-8", "replace"))
        return BwtMergeSortResult(options.outfn, options.fn)
