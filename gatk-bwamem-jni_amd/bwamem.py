"""Host-side mirror of the reference's Java API over the C ABI of libbwamem_hip.so.

The reference host is Java (src/main/java/org/broadinstitute/hellbender/utils/bwa/); no JDK
exists in this build environment, so the same operator interface is mirrored here -- same
class and method names, argument meaning and error behaviour -- so that tests read like
BwaMemIndexTest.java.  The Java classes themselves stay unchanged and bind the very same
library through the JNI glue.

  BwaMemIndex          <- BwaMemIndex.java:29-488
  BwaMemAligner        <- BwaMemAligner.java:20-327
  BwaMemAlignment      <- BwaMemAlignment.java:8-65
  BwaMemPairEndStats   <- BwaMemPairEndStats.java:15-205

The library is the product: if libbwamem_hip.so is missing this module fails to import.
"""
import ctypes
import math
import os
import struct
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("LIBBWA_PATH", os.path.join(_HERE, "libbwamem_hip.so"))   # BwaMemIndex.java:438-441
if not os.path.exists(_LIB_PATH):
    raise ImportError("native library %s not found: build it with __graft_entry__.build()" % _LIB_PATH)
_lib = ctypes.CDLL(_LIB_PATH)

_lib.jnibwa_createReferenceIndex.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
_lib.jnibwa_createIndexFile.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
_lib.jnibwa_openIndex.restype = ctypes.c_void_p
_lib.jnibwa_openIndex.argtypes = [ctypes.c_int]
_lib.jnibwa_destroyIndex.argtypes = [ctypes.c_void_p]
_lib.jnibwa_getRefContigNames.restype = ctypes.c_void_p
_lib.jnibwa_getRefContigNames.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
_lib.jnibwa_createAlignments.restype = ctypes.c_void_p
_lib.jnibwa_createAlignments.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
_lib.jnibwa_createDefaultOptions.restype = ctypes.c_void_p
_lib.jnibwa_free.argtypes = [ctypes.c_void_p]
_lib.jnibwa_getVersion.restype = ctypes.c_char_p


class CouldNotReadImageException(RuntimeError):
    pass


class CouldNotCreateIndexException(RuntimeError):
    pass


class BwaMemPairEndStats:
    DEFAULT_LOW_AND_HIGH_SIGMA = 4
    DEFAULT_STD_TO_AVERAGE_RATIO = .1

    def __init__(self, average=None, std=None, low=None, high=None, _failed=False):
        if _failed:
            self.failed, self.average, self.std, self.low, self.high = True, float("nan"), float("nan"), 0, 0
            return
        if std is None:
            std = average * self.DEFAULT_STD_TO_AVERAGE_RATIO
        if low is None:
            low = max(1, int(math.floor(average - self.DEFAULT_LOW_AND_HIGH_SIGMA * std + 0.5)))
            high = max(1, int(math.floor(average + self.DEFAULT_LOW_AND_HIGH_SIGMA * std + 0.5)))
        if math.isnan(average) or math.isinf(average) or average < 1:
            raise ValueError("invalid input average: %r" % average)
        if math.isnan(std) or math.isinf(std) or std < 0:
            raise ValueError("invalid std. err: %r" % std)
        if low > average:
            raise ValueError("the low limit cannot be larger than the average")
        if high < average:
            raise ValueError("the high limit cannot be larger than the average")
        self.failed, self.average, self.std, self.low, self.high = False, float(average), float(std), int(low), int(high)

    def _pack(self):
        """mem_pestat_t[4] as ...BwaMemIndex.c:21-40 fills it: slot 1 (FR) from this object, the rest failed."""
        b = b""
        for i in range(4):
            if i == 1 and not self.failed:
                b += struct.pack("<iiiidd", self.low, self.high, 0, 0, self.average, self.std)
            else:
                b += struct.pack("<iiiidd", 0, 0, 1, 0, 0.0, 0.0)
        return b


BwaMemPairEndStats.FAILED = BwaMemPairEndStats(_failed=True)
BwaMemPairEndStats.DO_NOT_INFER = BwaMemPairEndStats.FAILED


class BwaMemAlignment:
    def __init__(self, samFlag, refId, refStart, refEnd, seqStart, seqEnd, mapQual, nMismatches, alignerScore,
                 suboptimalScore, cigar, mdTag, xaTag, mateRefId, mateRefStart, templateLen):
        self.samFlag, self.refId, self.refStart, self.refEnd = samFlag, refId, refStart, refEnd
        self.seqStart, self.seqEnd, self.mapQual, self.nMismatches = seqStart, seqEnd, mapQual, nMismatches
        self.alignerScore, self.suboptimalScore, self.cigar, self.mdTag, self.xaTag = alignerScore, suboptimalScore, cigar, mdTag, xaTag
        self.mateRefId, self.mateRefStart, self.templateLen = mateRefId, mateRefStart, templateLen

    def getSamFlag(self): return self.samFlag
    def getRefId(self): return self.refId
    def getRefStart(self): return self.refStart
    def getRefEnd(self): return self.refEnd
    def getSeqStart(self): return self.seqStart
    def getSeqEnd(self): return self.seqEnd
    def getMapQual(self): return self.mapQual
    def getNMismatches(self): return self.nMismatches
    def getAlignerScore(self): return self.alignerScore
    def getSuboptimalScore(self): return self.suboptimalScore
    def getCigar(self): return self.cigar
    def getMDTag(self): return self.mdTag
    def getXATag(self): return self.xaTag
    def getMateRefId(self): return self.mateRefId
    def getMateRefStart(self): return self.mateRefStart
    def getTemplateLen(self): return self.templateLen


class BwaMemIndex:
    INDEX_FILE_EXTENSIONS = [".amb", ".ann", ".bwt", ".pac", ".sa"]
    IMAGE_FILE_EXTENSION = ".img"
    _class_lock = threading.Lock()

    @staticmethod
    def createIndexImageFromIndexFiles(indexPrefix, imageFile):
        if indexPrefix is None:
            raise ValueError("the index prefix cannot be null")
        if imageFile is None:
            raise ValueError("the image file cannot be null")
        for ext in BwaMemIndex.INDEX_FILE_EXTENSIONS:       # assertLooksLikeIndexPrefix, BwaMemIndex.java:259-266
            fn = indexPrefix + ext
            if not (os.path.isfile(fn) and os.path.getsize(fn) > 0):
                raise ValueError("index file %s is missing or empty" % fn)
        _lib.jnibwa_createIndexFile(indexPrefix.encode(), imageFile.encode())   # result discarded, BwaMemIndex.java:122

    @staticmethod
    def createIndexImageFromFastaFile(fasta, imageFile=None, algo="auto"):
        if imageFile is None:
            imageFile = fasta + BwaMemIndex.IMAGE_FILE_EXTENSION
        if not (os.path.isfile(fasta) and os.path.getsize(fasta) > 0):
            raise ValueError("the fasta file %s is missing or empty" % fasta)
        import tempfile
        tmp = tempfile.mkdtemp()
        prefix = os.path.join(tmp, os.path.basename(fasta))
        try:
            rc = _lib.jnibwa_createReferenceIndex(fasta.encode(), prefix.encode(), algo.encode())
            if rc == -1:
                raise ValueError("wrong algorithm name '%s'" % algo)     # ...BwaMemIndex.c:52-58
            _lib.jnibwa_createIndexFile(prefix.encode(), imageFile.encode())
        finally:
            for ext in BwaMemIndex.INDEX_FILE_EXTENSIONS:
                try:
                    os.unlink(prefix + ext)
                except OSError:
                    pass
            os.rmdir(tmp)
        return imageFile

    def __init__(self, indexImageFile):
        self.indexImageFile = indexImageFile
        if not (os.path.isfile(indexImageFile) and os.path.getsize(indexImageFile) > 0):
            raise CouldNotReadImageException("%s is empty or is not readable" % indexImageFile)
        self._refCount = 0
        self._lock = threading.Lock()
        fd = os.open(indexImageFile, os.O_RDONLY)          # ...BwaMemIndex.c:74-81
        self.indexAddress = _lib.jnibwa_openIndex(fd) or 0
        if not self.indexAddress:
            raise CouldNotReadImageException("%s: unable to open bwa-mem index" % indexImageFile)
        sz = ctypes.c_size_t()
        p = _lib.jnibwa_getRefContigNames(self.indexAddress, ctypes.byref(sz))
        if not p:
            raise CouldNotReadImageException("unable to retrieve reference contig names from bwa-mem index")
        buf = ctypes.string_at(p, sz.value)
        _lib.jnibwa_free(p)
        n, = struct.unpack_from("=i", buf, 0)
        off, self.refContigNames = 4, []
        for _ in range(n):
            l, = struct.unpack_from("=i", buf, off); off += 4
            self.refContigNames.append(buf[off:off + l].decode()); off += l

    def isOpen(self):
        return self.indexAddress != 0

    def refIndex(self):
        with self._lock:
            self._refCount += 1
        if not self.indexAddress:
            raise RuntimeError("Index image %s has been closed" % self.indexImageFile)
        return self.indexAddress

    def deRefIndex(self):
        with self._lock:
            self._refCount -= 1

    def close(self):
        if self.indexAddress:
            with BwaMemIndex._class_lock:
                if self.indexAddress:
                    if self._refCount != 0:
                        raise RuntimeError("Index image %s can't be closed:  it's in use." % self.indexImageFile)
                    addr, self.indexAddress = self.indexAddress, 0
                    _lib.jnibwa_destroyIndex(addr)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def getReferenceContigNames(self):
        return self.refContigNames

    @staticmethod
    def getBWAVersion():
        return _lib.jnibwa_getVersion().decode()

    def doAlignment(self, seqs, opts, peStats):
        sz = ctypes.c_size_t()
        pb = ctypes.create_string_buffer(peStats._pack(), 128) if peStats is not None else None
        p = _lib.jnibwa_createAlignments(self.indexAddress, opts, pb, seqs, ctypes.byref(sz))
        if not p:
            raise RuntimeError("Unable to get alignments from bwa-mem index %s: We don't know why." % self.indexImageFile)
        out = ctypes.string_at(p, sz.value)
        _lib.jnibwa_free(p)
        return out


def _opt_int(off):
    return (lambda self: struct.unpack_from("=i", self._getOpts(), off)[0],
            lambda self, v: struct.pack_into("=i", self._getOpts(), off, v))


def _opt_float(off):
    return (lambda self: struct.unpack_from("=f", self._getOpts(), off)[0],
            lambda self, v: struct.pack_into("=f", self._getOpts(), off, v))


class BwaMemAligner:
    MEM_F_PE, MEM_F_NOPAIRING, MEM_F_ALL, MEM_F_NO_MULTI, MEM_F_NO_RESCUE = 0x2, 0x4, 0x8, 0x10, 0x20
    MEM_F_REF_HDR, MEM_F_SOFTCLIP, MEM_F_SMARTPE, MEM_F_PRIMARY5 = 0x100, 0x200, 0x400, 0x800

    def __init__(self, index):
        self.index = index
        if not index.isOpen():
            raise RuntimeError("Can't create aligner: bwa-mem index has been closed")
        p = _lib.jnibwa_createDefaultOptions()
        self.opts = (ctypes.c_char * 168).from_buffer_copy(ctypes.string_at(p, 168))
        _lib.jnibwa_free(p)
        self.pairEndStats = None

    def isOpen(self):
        return self.opts is not None

    def close(self):
        self.opts = None

    def _getOpts(self):
        if self.opts is None:
            raise RuntimeError("The aligner has been closed.")
        return self.opts

    # option accessors by byte offset, BwaMemAligner.java:46-136
    getMatchScoreOption, setMatchScoreOption = _opt_int(0)
    getMismatchPenaltyOption, setMismatchPenaltyOption = _opt_int(4)
    getDGapOpenPenaltyOption, setDGapOpenPenaltyOption = _opt_int(8)
    getDGapExtendPenaltyOption, setDGapExtendPenaltyOption = _opt_int(12)
    getIGapOpenPenaltyOption, setIGapOpenPenaltyOption = _opt_int(16)
    getIGapExtendPenaltyOption, setIGapExtendPenaltyOption = _opt_int(20)
    getUnpairedPenaltyOption, setUnpairedPenaltyOption = _opt_int(24)
    getClip5PenaltyOption, setClip5PenaltyOption = _opt_int(28)
    getClip3PenaltyOption, setClip3PenaltyOption = _opt_int(32)
    getBandwidthOption, setBandwidthOption = _opt_int(36)
    getZDropOption, setZDropOption = _opt_int(40)
    getOutputScoreThresholdOption, setOutputScoreThresholdOption = _opt_int(56)
    getFlagOption, setFlagOption = _opt_int(60)
    getMinSeedLengthOption, setMinSeedLengthOption = _opt_int(64)
    getMinChainWeightOption, setMinChainWeightOption = _opt_int(68)
    getMaxChainExtendOption, setMaxChainExtendOption = _opt_int(72)
    getSplitFactorOption, setSplitFactorOption = _opt_float(76)
    getSplitWidthOption, setSplitWidthOption = _opt_int(80)
    getMaxSeedOccurencesOption, setMaxSeedOccurencesOption = _opt_int(84)
    getMaxChainGapOption, setMaxChainGapOption = _opt_int(88)
    getNThreadsOption, setNThreadsOption = _opt_int(92)
    getChunkSizeOption, setChunkSizeOption = _opt_int(96)
    getMaskLevelOption, setMaxLevelOption = _opt_float(100)
    getDropRatioOption, setDropRatioOption = _opt_float(104)
    getXADropRatio, setXADropRatio = _opt_float(108)
    getMaskLevelRedunOption, setMaskLevelRedunOption = _opt_float(112)
    getMapQCoefLenOption, setMapQCoefLenOption = _opt_float(116)
    getMapQCoefFacOption, setMapQCoefFacOption = _opt_int(120)
    getMaxInsOption, setMaxInsOption = _opt_int(124)
    getMaxMateSWOption, setMaxMateSWOption = _opt_int(128)
    getMaxXAHitsOption, setMaxXAHitsOption = _opt_int(132)
    getMaxXAHitsAltOption, setMaxXAHitsAltOption = _opt_int(136)

    def getMaxMemIntvOption(self): return struct.unpack_from("=q", self._getOpts(), 48)[0]
    def setMaxMemIntvOption(self, v): struct.pack_into("=q", self._getOpts(), 48, v)
    def getScoringMatrixOption(self): return bytes(self._getOpts()[140:165])
    def setScoringMatrixOption(self, mat): self._getOpts()[140:165] = bytes(x & 0xff for x in mat)
    def getExpectedOptsSize(self): return 168
    def getOptsSize(self): return len(self._getOpts())

    def alignPairs(self): self.setFlagOption(self.MEM_F_PE | self.getFlagOption())

    def setIntraCtgOptions(self):
        self.setDGapOpenPenaltyOption(16); self.setIGapOpenPenaltyOption(16); self.setMismatchPenaltyOption(9)
        self.setClip5PenaltyOption(5); self.setClip3PenaltyOption(5)

    def inferPairEndStats(self): self.pairEndStats = None
    def dontInferPairEndStats(self): self.pairEndStats = BwaMemPairEndStats.DO_NOT_INFER
    def setProperPairEndStats(self, stats): self.pairEndStats = stats
    def getIndex(self): return self.index

    def alignSeqsRaw(self, sequences, func=lambda s: s):
        """the raw native response (BwaMemAligner.java:192-214)"""
        opts = self._getOpts()
        self.index.refIndex()
        try:
            seqs = [func(e) for e in sequences]
            seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
            buf = struct.pack("=i", len(seqs)) + b"".join(s + b"\0" for s in seqs)
            contigBuf = ctypes.create_string_buffer(buf, len(buf))
            self._n = len(seqs)
            return self.index.doAlignment(contigBuf, opts, self.pairEndStats)
        finally:
            self.index.deRefIndex()

    def alignSeqs(self, sequences, func=lambda s: s):
        """BwaMemAligner.java:192-310: one list of BwaMemAlignment per input sequence"""
        buf = self.alignSeqsRaw(sequences, func)
        nSequences, off, out = self._n, 0, []
        ops = "MID?S???????????"
        for _ in range(nSequences):
            nAligns, = struct.unpack_from("=i", buf, off); off += 4
            alignments = []
            for _ in range(nAligns):
                flag_mapQ, = struct.unpack_from("=i", buf, off); off += 4
                flags, mapQual = (flag_mapQ >> 16) & 0xffff, flag_mapQ & 0xff
                if flags & 0x4:
                    refId = refStart = refEnd = seqStart = seqEnd = -1
                    nMismatches = alignerScore = suboptimalScore = 0
                    cigar, mdTag, xaTag = "", None, None
                else:
                    refId, refStart, nMismatches, alignerScore, suboptimalScore, nCigarOps = struct.unpack_from("=6i", buf, off); off += 24
                    cigar, refLen, seqLen, seqStart = "", 0, 0, 0
                    for i in range(max(nCigarOps, 0)):
                        lenOp, = struct.unpack_from("=I", buf, off); off += 4
                        ln, op = lenOp >> 4, ops[lenOp & 0xf]
                        cigar += "%d%s" % (ln, op)
                        if i == 0 and op == "S": seqStart = ln
                        if op in "MD": refLen += ln
                        if op in "MI": seqLen += ln
                    refEnd, seqEnd = refStart + refLen, seqStart + seqLen
                    tags = []
                    for _t in range(2):
                        tagLen, = struct.unpack_from("=i", buf, off); off += 4
                        tags.append(buf[off:off + tagLen].decode() if tagLen else None)
                        off += (tagLen + 3) & ~3
                    mdTag, xaTag = tags
                if (flags & 0x1) == 0 or (flags & 0x8) != 0:
                    mateRefId, mateStartPos, templateLen = -1, -1, 0
                else:
                    mateRefId, mateStartPos, templateLen = struct.unpack_from("=3i", buf, off); off += 12
                alignments.append(BwaMemAlignment(flags, refId, refStart, refEnd, seqStart, seqEnd, mapQual, nMismatches,
                                                  alignerScore, suboptimalScore, cigar, mdTag, xaTag, mateRefId, mateStartPos, templateLen))
            out.append(alignments)
        return out
