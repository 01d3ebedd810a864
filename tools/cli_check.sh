# Manual GPU smoke run of the CLIs with every model family (run through gpurun from the repo root).
set -e
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT/learn-nerf_amd:$GRAFT_REPO_ROOT
D=/tmp/cube_ds
rm -rf $D /tmp/ckpt_*.pkl
python learn-nerf_amd/learn_nerf/scripts/make_cube_dataset.py --views 12 --size 32 $D > /dev/null
python learn-nerf_amd/learn_nerf/scripts/train_nerf.py --instant_ngp --seed 1 --lr 1e-2 --batch_size 512 --coarse_samples 16 --fine_samples 32 --max_steps 60 --save_path /tmp/ckpt_ngp.pkl $D | tail -2
python learn-nerf_amd/learn_nerf/scripts/train_nerf.py --instant_ngp --ref_nerf --seed 1 --lr 1e-2 --batch_size 256 --coarse_samples 16 --fine_samples 32 --max_steps 20 --save_path /tmp/ckpt_ngpref.pkl $D | tail -1
python learn-nerf_amd/learn_nerf/scripts/train_nerf.py --ref_nerf --seed 1 --lr 1e-3 --batch_size 256 --coarse_samples 16 --fine_samples 32 --max_steps 20 --save_path /tmp/ckpt_ref.pkl $D | tail -1
python learn-nerf_amd/learn_nerf/scripts/train_nerf.py --ref_nerf --precision fp32 --seed 1 --lr 1e-3 --batch_size 256 --coarse_samples 16 --fine_samples 32 --max_steps 5 --save_path /tmp/ckpt_ref32.pkl $D | tail -1
python learn-nerf_amd/learn_nerf/scripts/render_nerf.py --instant_ngp --seed 0 --batch_size 512 --coarse_samples 16 --fine_samples 32 --width 16 --height 16 --model_path /tmp/ckpt_ngp.pkl $D/metadata.json $D/0000.json /tmp/ngp_render.png && python -c "from PIL import Image; import numpy as np; print('render', np.array(Image.open('/tmp/ngp_render.png')).shape)"
