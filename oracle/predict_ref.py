"""CPU restatement of the reference's `src/predict.py` end-to-end flow -- TEST INFRASTRUCTURE ONLY (same rules as
dfa_oracle.py: used by tests/ and by the `cpu_baseline.end_to_end` leg of bench.py, never by the product).

features.pkl -> pandas unpickle -> FeatureOnlyDataset (`.iloc[idx].float()` per item, src/predict.py:55-63) ->
DataLoader(batch_size=32, num_workers=2, shuffle=False) (:92-98) -> eval forward of the stock PyTorch CPU layer stack
(oracle/torch_ref.py restating src/model.py:33-42) on `features.transpose(1, 2)` (:104-106) -> sigmoid (:107-108) ->
python floats (:111) -> DataFrame{uttid, predictions} -> to_pickle (:116-122)."""
from __future__ import annotations

import pandas as pd
import torch
from torch.utils.data import DataLoader, Dataset

from . import torch_ref as R


class FeatureOnlyDataset(Dataset):
    def __init__(self, features_df):
        self.features = features_df["features"].reset_index(drop=True)

    def __len__(self):
        return len(self.features)

    def __getitem__(self, idx):
        return self.features.iloc[idx].float()


def predict_end_to_end(features_path, sd, out_path, batch_size=32, num_workers=2, apply_sigmoid=True, swap_tf=True):
    features_df = pd.read_pickle(features_path)
    if "uttid" not in features_df.columns:
        raise ValueError("features.pkl must contain 'uttid'")
    loader = DataLoader(FeatureOnlyDataset(features_df), batch_size=batch_size, num_workers=num_workers, shuffle=False)
    predictions = []
    with torch.no_grad():
        for features in loader:
            if swap_tf:
                features = features.transpose(1, 2)
            logits = R.cnn2d_forward(sd, features).squeeze(-1)
            scores = torch.sigmoid(logits) if apply_sigmoid else logits
            predictions.extend(scores.detach().cpu().tolist())
    if len(predictions) != len(features_df):
        raise ValueError("Number of predictions does not match number of rows in features.pkl")
    pred_df = pd.DataFrame({"uttid": features_df["uttid"].values, "predictions": predictions})
    pred_df.to_pickle(out_path)
    return pred_df
