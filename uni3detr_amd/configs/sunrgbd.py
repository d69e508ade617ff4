"""Model section of the SUN RGB-D Uni3DETR configuration, restated so that bench.py / smoke() can build the flagship model on
the GPU box, where the reference tree does not exist.  tests/test_plugin_cpu.py asserts it equals the `model` dict of the
shipped projects/configs/uni3detr/uni3detr_sunrgbd.py (:26-140) whenever that file is available."""
RANGE = [-3.2, -0.2, -2., 3.2, 6.2, 0.56]
VOXEL = [0.02, 0.02, 0.02]
GRID = [128, 320, 320]
AMP = True

_layer = dict(
    type='BaseTransformerLayer',
    attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1),
               dict(type='UniCrossAtten', num_points=1, embed_dims=256, num_sweeps=1, fp16_enabled=AMP)],
    ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=512, num_fcs=2, ffn_drop=0.1, act_cfg=dict(type='ReLU', inplace=True)),
    norm_cfg=dict(type='LN'),
    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'))

_bn3 = dict(type='BN3d', eps=1e-3, momentum=0.01)

model = dict(
    type='Uni3DETR',
    pts_voxel_layer=dict(max_num_points=5, voxel_size=VOXEL, max_voxels=(16000, 40000), point_cloud_range=RANGE),
    pts_voxel_encoder=dict(type='HardSimpleVFE', num_features=4),
    pts_middle_encoder=dict(type='SparseEncoderHD', in_channels=4, sparse_shape=GRID, output_channels=256, order=('conv', 'norm', 'act'),
                            encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                            encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock', fp16_enabled=False),
    pts_backbone=dict(type='SECOND3D', in_channels=[256, 256, 256], out_channels=[128, 256, 512], layer_nums=[5, 5, 5],
                      layer_strides=[1, 2, 4], is_cascade=False, norm_cfg=dict(_bn3), conv_cfg=dict(type='Conv3d', kernel=(1, 3, 3), bias=False)),
    pts_neck=dict(type='SECOND3DFPN', in_channels=[128, 256, 512], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                  norm_cfg=dict(_bn3), upsample_cfg=dict(type='deconv3d', bias=False),
                  extra_conv=dict(type='Conv3d', num_conv=3, bias=False), use_conv_for_no_stride=True),
    pts_bbox_head=dict(
        type='Uni3DETRHead', num_query=300, num_classes=10, in_channels=256, sync_cls_avg_factor=True, with_box_refine=True,
        as_two_stage=False, code_size=8,
        transformer=dict(type='Uni3DETRTransformer', fp16_enabled=AMP,
                         decoder=dict(type='Uni3DETRTransformerDecoder', num_layers=3, return_intermediate=True, transformerlayers=_layer)),
        bbox_coder=dict(type='NMSFreeCoder', post_center_range=RANGE, pc_range=RANGE, max_num=1000, voxel_size=VOXEL, alpha=1.0, num_classes=10),
        post_processing=dict(type='nms', nms_thr=0.5),
        positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True, offset=-0.5),
        loss_cls=dict(type='SoftFocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.5),
        loss_bbox=dict(type='L1Loss', loss_weight=0.25),
        loss_iou=dict(type='IoU3DLoss', loss_weight=1.2),
        code_weights=[1.0] * 8),
    train_cfg=dict(pts=dict(grid_size=GRID, voxel_size=VOXEL, point_cloud_range=RANGE, out_size_factor=4,
                            assigner=dict(type='HungarianAssigner3D', cls_cost=dict(type='FocalLossCost', weight=2.0),
                                          reg_cost=dict(type='BBox3DL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=1.2),
                                          pc_range=RANGE))))
