"""BwtMergeSortOptions / ArgInfo mirror.

Same option names, defaults and grammar as the reference CLI
(/root/reference/src/bwtb3m.cpp:29-56, /root/reference/README.md:31-50): leading key=value
tokens are options, the remaining tokens are positional; numeric values accept k/m/g suffixes.
"""
import os
import socket
import time

from ._lib import B3MError, INPUT_TYPES


def parse_numeric(v):
    """getValueUnsignedNumeric: decimal digits with an optional k/m/g(/t) suffix (x1024)"""
    s = str(v)
    i = 0
    while i < len(s) and s[i].isdigit():
        i += 1
    if i == 0:
        raise B3MError("cannot parse numeric value " + s)
    x = int(s[:i])
    rest = s[i:].lower()
    if rest:
        mul = {"k": 1024, "m": 1024 ** 2, "g": 1024 ** 3, "t": 1024 ** 4}.get(rest)
        if mul is None:
            raise B3MError("cannot parse numeric value " + s)
        x *= mul
    return x


class ArgInfo:
    """libmaus2::util::ArgInfo: argv -> (argmap, restargs)"""

    def __init__(self, argv):
        self.progname = argv[0] if argv else ""
        self.argmap = {}
        self.restargs = []
        self.help = False
        i = 1
        while i < len(argv):
            a = argv[i]
            if a in ("-h", "--help"):
                self.help = True
                i += 1
                continue
            eq = a.find("=")
            if eq <= 0:
                break
            self.argmap[a[:eq]] = a[eq + 1:]
            i += 1
        self.restargs = list(argv[i:])

    def helpRequested(self):
        return self.help

    def getDefaultTmpFileName(self):
        # <prog>_<host>_<pid>_<time>  (/root/reference/README.md:39,45)
        return "%s_%s_%d_%d" % (os.path.basename(self.progname), socket.gethostname(), os.getpid(), int(time.time()))


class BwtMergeSortOptions:
    """option struct of libmaus2::suffixsort::bwtb3m::BwtMergeSortOptions"""

    @staticmethod
    def getDefaultSaSamplingRate():
        return 32

    @staticmethod
    def getDefaultIsaSamplingRate():
        return 262144

    @staticmethod
    def getDefaultMem():
        return 0  # reference: 2 GiB of host RAM; this engine: 0 = all free HBM

    @staticmethod
    def getDefaultNumThreads():
        return os.cpu_count() or 1

    @staticmethod
    def getDefaultBWTOnly():
        return 0

    @staticmethod
    def getDefaultCopyInputToMemory():
        return 0

    @staticmethod
    def getDefaultLargeLCPThres():
        return 16384

    @staticmethod
    def getDefaultVerbose():
        return 0

    @staticmethod
    def getDefaultInputType():
        return "bytestream"

    @staticmethod
    def parseInputType(name):
        if name not in INPUT_TYPES:
            # message of the reference's switch, /root/reference/src/checkbwt.cpp:274
            raise B3MError("libmaus2::suffixsort::bwtb3m::BwtMergeSort::computeBwt: unknown/unsupported input type")   # /root/reference/src/checkbwt.cpp:274
        return INPUT_TYPES[name]

    def __init__(self, arginfo=None, **kw):
        am = dict(arginfo.argmap) if arginfo is not None else {}
        am.update({k: str(v) for k, v in kw.items()})
        tmp = arginfo.getDefaultTmpFileName() if arginfo is not None else ArgInfo(["bwtb3m"]).getDefaultTmpFileName()
        self.fn = arginfo.restargs[0] if (arginfo is not None and arginfo.restargs) else am.get("fn", "")
        self.inputtype = am.get("inputtype", self.getDefaultInputType())
        self.parseInputType(self.inputtype)
        self.tmpfilenamebase = am.get("tmpprefix", tmp)
        self.sparsetmpfilenamebase = am.get("sparsetmpprefix", self.tmpfilenamebase)
        self.outfn = am.get("outputfilename", self.tmpfilenamebase + ".bwt")
        self.bwtonly = parse_numeric(am.get("bwtonly", self.getDefaultBWTOnly()))
        self.mem = parse_numeric(am.get("mem", self.getDefaultMem()))
        self.sasamplingrate = parse_numeric(am.get("sasamplingrate", self.getDefaultSaSamplingRate()))
        # 262144, or 64 when bwtonly=1 (/root/reference/ChangeLog:281)
        self.isasamplingrate = parse_numeric(am.get("isasamplingrate", 64 if self.bwtonly else self.getDefaultIsaSamplingRate()))
        self.numthreads = parse_numeric(am.get("numthreads", self.getDefaultNumThreads()))
        self.copyinputtomemory = parse_numeric(am.get("copyinputtomemory", self.getDefaultCopyInputToMemory()))
        self.largelcpthres = parse_numeric(am.get("largelcpthres", self.getDefaultLargeLCPThres()))
        self.verbose = parse_numeric(am.get("verbose", self.getDefaultVerbose()))
        self.blocksize = parse_numeric(am.get("blocksize", 0))  # extension
        self.numgpus = parse_numeric(am.get("numgpus", 1))      # extension: GPUs of this node (one host thread + one suffix-range shard each; b3m_options.num_gpus)
