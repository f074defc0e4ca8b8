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


class DeviceHandposeEvaluation(HandposeEvaluation):
    """The same numeric interface with the arithmetic on the device (dpp_pose_eval, csrc/prior.hip): one launch pair produces the
    per-joint errors, the per-frame mean / max / std, the per-joint statistics and the frames-within-distance counts for a
    threshold list; the getters read the cached result (thresholds not in the cached list trigger another launch)."""

    def __init__(self, gt, joints, dolegend=True, linewidth=1, runtime=None, thresholds=None):
        super(DeviceHandposeEvaluation, self).__init__(gt, joints, dolegend, linewidth)
        from hipdp.runtime import default_runtime
        self.rt = runtime or default_runtime()
        self._gt_dev = self.rt.upload(numpy.ascontiguousarray(self.gt, numpy.float32))
        self._pr_dev = self.rt.upload(numpy.ascontiguousarray(self.joints, numpy.float32))
        self._thr = numpy.asarray(list(range(0, 81)) if thresholds is None else thresholds, numpy.float64)
        self._res = None

    def _run(self):
        if self._res is not None:
            return self._res
        from hipdp.lib import check
        rt = self.rt
        N, J = self.gt.shape[0], self.gt.shape[1]
        T = len(self._thr)
        thr = rt.upload(self._thr)
        err, frame, out = rt.alloc((N, J), numpy.float64), rt.alloc((N, 4), numpy.float64), rt.alloc(4 + 3 * J + 2 * T, numpy.float64)
        check(rt.lib.dpp_pose_eval(self._gt_dev.ptr, self._pr_dev.ptr, N, J, thr.ptr, T, err.ptr, frame.ptr, out.ptr, rt.stream), 'dpp_pose_eval')
        rt.synchronize()
        o = out.get()
        self._res = dict(err=err.get(), frame=frame.get(), mean=o[0], max=o[1], std=o[2], jmean=o[4:4 + J], jstd=o[4 + J:4 + 2 * J],
                         jmax=o[4 + 2 * J:4 + 3 * J], within_max=o[4 + 3 * J:4 + 3 * J + T], within_mean=o[4 + 3 * J + T:4 + 3 * J + 2 * T])
        return self._res

    def _err(self):
        return self._run()['err']

    def getMeanError(self):
        return float(self._run()['mean'])

    def getStdError(self):
        return float(self._run()['std'])

    def getMeanErrorOverSeq(self):
        return self._run()['frame'][:, 0]

    def getMaxError(self):
        return float(self._run()['max'])

    def getMaxErrorOverSeq(self):
        return self._run()['frame'][:, 1]

    def getJointMeanError(self, jointID):
        return float(self._run()['jmean'][jointID])

    def getJointStdError(self, jointID):
        return float(self._run()['jstd'][jointID])

    def getJointMaxError(self, jointID):
        return float(self._run()['jmax'][jointID])

    def _within(self, dist, key):
        idx = numpy.nonzero(self._thr == float(dist))[0]
        if len(idx) == 0:
            self._thr = numpy.append(self._thr, float(dist))
            self._res = None
            idx = [len(self._thr) - 1]
        return int(self._run()[key][idx[0]])

    def getNumFramesWithinMaxDist(self, dist):
        return self._within(dist, 'within_max')

    def getNumFramesWithinMeanDist(self, dist):
        return self._within(dist, 'within_mean')


class ICVLHandposeEvaluation(HandposeEvaluation):
    pass


class MSRAHandposeEvaluation(HandposeEvaluation):
    pass


class NYUHandposeEvaluation(HandposeEvaluation):
    pass
