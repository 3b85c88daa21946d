"""MVP data path: shard reader, mirror preprocessing and result writers / gatherer
(reference pointnet2/mvp_dataloader/mvp_dataset.py, data_utils/mirror_partial.py,
generate_samples_distributed.py:26-97)."""
