"""Post-training metrics on denormalised outputs (host side, numpy/scipy) — same quantities and
accumulation rule as the reference's ModelMetric (src/cae_tools/models/model_metric.py:25-71):
masked pixels of every instance are pooled for mse / rmse / mae; Pearson r is computed per
instance and averaged.  This is reporting, not part of the GPU hot path."""
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
