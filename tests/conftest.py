import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# Collection order of the GPU tier (round-5 review, item 1): deterministic kernel cases first, then the box side, the models, the in-situ shadows, the
# distributed cases, and the statistical admission gates LAST -- with `-x` a gate can then never hide a deterministic case behind it.  Files not named here keep
# their alphabetical place between the models and the in-situ group; the CPU tier is left in collection order.
_GPU_ORDER = [
    'test_gpu_kernels.py', 'test_gpu_tf_known_answers.py', 'test_gpu_augment.py',                                  # kernels against the oracle / known answers
    'test_gpu_retina.py', 'test_gpu_dense_heads.py', 'test_gpu_refinedet.py', 'test_gpu_lhrcnn.py',                # box side: matching, losses, decode, NMS
    'test_gpu_ssd300.py', 'test_gpu_ssd300_b32.py', 'test_gpu_ssd512.py', 'test_gpu_retinanet_model.py',           # whole models
    'test_gpu_yolov3.py', 'test_gpu_yolov2.py', 'test_gpu_fcos_model.py', 'test_gpu_centernet_model.py',
    'test_gpu_refinedet_model.py', 'test_gpu_pfpnet_model.py', 'test_gpu_engine_bf16.py',
    None,                                                                                                          # (anything else)
    'test_gpu_insitu_configs.py', 'test_gpu_dist.py', 'test_gpu_bf16_gate.py',
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_GPU_ORDER)}
    other = rank[None]

    def key(it):
        if it.get_closest_marker('gpu') is None:
            return (-1, 0)
        return (rank.get(os.path.basename(str(it.fspath)), other), 0)
    items.sort(key=key)            # stable: the order inside a file is kept
