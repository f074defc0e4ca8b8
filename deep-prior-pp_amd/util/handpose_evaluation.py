"""HandposeEvaluation numeric metrics (API of /root/reference/src/util/handpose_evaluation.py:49-228): mean / max /
per-joint Euclidean error in mm.  The matplotlib / VTK plotting of the reference is out of scope."""
import numpy


class HandposeEvaluation(object):
    def __init__(self, gt, joints, dolegend=True, linewidth=1):
        if not (isinstance(gt, numpy.ndarray) or isinstance(gt, list)) or not (isinstance(joints, list) or isinstance(joints, numpy.ndarray)):
            raise ValueError("Params must be list or ndarray")
        if len(gt) != len(joints):
            print("Error: groundtruth has {} elements, eval data has {}".format(len(gt), len(joints)))
            raise ValueError("Params must be the same size")
        if len(gt) == len(joints) == 0:
            print("Error: groundtruth has {} elements, eval data has {}".format(len(gt), len(joints)))
            raise ValueError("Params must be of non-zero size")
        if gt[0].shape != joints[0].shape:
            print("Error: groundtruth has {} dims, eval data has {}".format(gt[0].shape, joints[0].shape))
            raise ValueError("Params must be of same dimensionality")
        self.gt = numpy.asarray(gt)
        self.joints = numpy.asarray(joints)
        assert self.gt.shape == self.joints.shape
        self.subfolder = './eval/'
        self.dolegend, self.linewidth = dolegend, linewidth

    @property
    def gtjoints(self):
        return self.gt

    def _err(self):
        return numpy.sqrt(numpy.square(self.gt - self.joints).sum(axis=2))

    def getMeanError(self):
        return numpy.nanmean(numpy.nanmean(self._err(), axis=1))

    def getStdError(self):
        return numpy.nanmean(numpy.nanstd(self._err(), axis=1))

    def getMeanErrorOverSeq(self):
        return numpy.nanmean(self._err(), axis=1)

    def getMedianError(self):
        return numpy.nanmedian(self._err())

    def getMaxError(self):
        return numpy.nanmax(self._err())

    def getMaxErrorOverSeq(self):
        return numpy.nanmax(self._err(), axis=1)

    def getJointMeanError(self, jointID):
        return numpy.nanmean(numpy.sqrt(numpy.square(self.gt[:, jointID, :] - self.joints[:, jointID, :]).sum(axis=1)))

    def getJointStdError(self, jointID):
        return numpy.nanstd(numpy.sqrt(numpy.square(self.gt[:, jointID, :] - self.joints[:, jointID, :]).sum(axis=1)))

    def getJointMaxError(self, jointID):
        return numpy.nanmax(numpy.sqrt(numpy.square(self.gt[:, jointID, :] - self.joints[:, jointID, :]).sum(axis=1)))

    def getNumFramesWithinMaxDist(self, dist):
        return (numpy.nanmax(self._err(), axis=1) <= dist).sum()

    def getNumFramesWithinMeanDist(self, dist):
        return (numpy.nanmean(self._err(), axis=1) <= dist).sum()

    def plotEvaluation(self, *args, **kwargs):
        raise NotImplementedError("plotting is out of scope for the hot path (SURVEY.md section 2, OOS)")

    plotResult = plotEvaluation


class ICVLHandposeEvaluation(HandposeEvaluation):
    pass


class MSRAHandposeEvaluation(HandposeEvaluation):
    pass


class NYUHandposeEvaluation(HandposeEvaluation):
    pass
