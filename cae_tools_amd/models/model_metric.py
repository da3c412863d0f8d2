"""Post-training metrics on denormalised outputs — same quantities and accumulation rule as the
reference's ModelMetric (src/cae_tools/models/model_metric.py:25-71): masked pixels of every
instance are pooled for mse / rmse / mae; Pearson r is computed per instance and averaged.

`ModelMetric` is the reference's host class (numpy arrays in).  `DeviceModelMetric` is what
BaseModel.evaluate uses: scores, truth and mask stay on the GPU, cae_metric_sums (include/cae_hip.h)
reduces every instance to eight fp64 sums and only those cross PCIe."""
import numpy as np


class ModelMetric:

    def __init__(self):
        self.actuals = []
        self.estimates = []

    def accumulate(self, actual, estimates, mask):
        actual = np.asarray(actual)
        estimates = np.asarray(estimates)
        if actual.shape != estimates.shape:
            raise ValueError("The shapes of 'actual' and 'estimates' must match.")
        keep = np.asarray(mask).reshape(-1).astype(bool)
        self.actuals.append(actual.reshape(-1)[keep])
        self.estimates.append(estimates.reshape(-1)[keep])

    def get_metrics(self):
        if not self.actuals or not self.estimates:
            raise ValueError("No data accumulated to calculate metrics.")
        from scipy.stats import pearsonr
        a = np.concatenate(self.actuals)
        e = np.concatenate(self.estimates)
        err = a - e
        mse = np.mean(err ** 2)
        r = [pearsonr(x, y)[0] for x, y in zip(self.actuals, self.estimates) if x.size and y.size]
        return {"mse": mse, "rmse": np.sqrt(mse), "mae": np.mean(np.abs(err)),
                "mean_pearson_correlation": np.mean(r) if r else 0.0}


def metrics_from_sums(sums):
    """(n_inst, 8) rows {n, Sa, Se, Saa, See, Sae, S|a-e|, S(a-e)^2} -> the reference's metric dict"""
    sums = np.asarray(sums, dtype=np.float64).reshape(-1, 8)
    if sums.shape[0] == 0:
        raise ValueError("No data accumulated to calculate metrics.")
    n_all = sums[:, 0].sum()
    mse = sums[:, 7].sum() / n_all     # numpy's mean of an empty pool is nan with a warning; so is 0/0 here
    mae = sums[:, 6].sum() / n_all
    rs = []
    for (n, sa, se, saa, see, sae, _, _) in sums:
        if n == 0:
            continue  # model_metric.py:60-61 skips empty instances
        cov = sae - sa * se / n
        va = saa - sa * sa / n
        ve = see - se * se / n
        with np.errstate(invalid="ignore", divide="ignore"):
            rs.append(np.clip(cov / np.sqrt(va * ve), -1.0, 1.0) if va > 0 and ve > 0 else np.nan)
    return {"mse": mse, "rmse": np.sqrt(mse), "mae": mae,
            "mean_pearson_correlation": np.mean(rs) if rs else 0.0}


class DeviceModelMetric:
    """ModelMetric over GPU-resident batches: accumulate(actual, y_normalised, mask, vmin, vmax)"""

    def __init__(self):
        self.sums = []

    def accumulate(self, actual, scores, mask, vmin, vmax):
        from .. import engine as _eng
        self.sums.append(_eng.metric_sums(scores, actual, mask, vmin, vmax))

    def get_metrics(self):
        if not self.sums:
            raise ValueError("No data accumulated to calculate metrics.")
        import torch
        return metrics_from_sums(torch.cat(self.sums).cpu().numpy())
