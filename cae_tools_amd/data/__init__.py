from .arrays import DataArray, Dataset, open_mfdataset, open_dataset  # noqa: F401
