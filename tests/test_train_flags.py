"""train.py keeps the reference's flag surface: names, types, defaults of all
flags of /root/reference/train.py:45-120 (golden extracted by tests/golden/make_flags_golden.py)
and the model_name / folder layout of train.py:133-166."""
import json
import os

import train

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_flags.json")))


def test_every_reference_flag_exists_with_same_default():
    opt = train.parse_option([])
    assert len(GOLD) == 49
    for flag, spec in GOLD.items():
        dest = flag.lstrip("-").replace("-", "_")
        assert hasattr(opt, dest), flag
        if flag == "--lr_decay_epochs":
            assert opt.lr_decay_epochs == [120, 160, 200]            # train.py:125-128
        elif "default" in spec:
            assert getattr(opt, dest) == spec["default"], flag
        elif spec.get("action") == "store_true":
            assert getattr(opt, dest) is False, flag


def test_flag_types_and_nargs():
    opt = train.parse_option(["--gpu", "3", "1", "--nce-k", "16384", "--moco", "--batch-size", "256",
                              "--learning_rate", "0.01", "--restart-prob", "0.5"])
    assert opt.gpu == [3, 1] and opt.nce_k == 16384 and opt.moco is True and opt.batch_size == 256
    assert opt.learning_rate == 0.01 and opt.restart_prob == 0.5


def test_model_name_matches_reference_format(tmp_path):
    opt = train.parse_option(["--exp", "Pretrain", "--moco", "--nce-k", "16384", "--model-path", str(tmp_path / "saved"),
                              "--tb-path", str(tmp_path / "tb")])
    opt = train.option_update(opt)
    # the name tests/utils.py:7 of the reference hard-codes for its MoCo checkpoint
    assert opt.model_name == ("Pretrain_moco_True_dgl_gin_layer_5_lr_0.005_decay_1e-05_bsz_32_hid_64_samples_2000_"
                              "nce_t_0.07_nce_k_16384_rw_hops_256_restart_prob_0.8_aug_1st_ft_False_deg_16_pos_32_"
                              "momentum_0.999")
    assert os.path.isdir(opt.model_folder) and os.path.isdir(opt.tb_folder)
