# Config of the 2D->3D lifting path in the reference's (mmcv python-config) format: the entries of
# projects/configs/verformer/vocc.py that the path reads -- model.pts_bbox_head (:87-195), model.train_cfg (:197-207),
# data.samples_per_gpu (:222), optimizer (:265-272), grad clip (:274) -- restated as data for MI355X runs where the
# reference tree (and its dataset / runtime _base_ files) is absent.  tests/test_config_cpu.py loads the reference's
# file through the same loader and asserts that these entries are equal.  Image backbone / neck / dataset / runner /
# hooks are outside the path (out of scope) and are not restated.
pc_range = [-6.0, -6.0, -1.5, 6.0, 6.0, 2.0]
grid = dict(z=4, h=15, w=15)                 # coarse voxel queries; the head refines to 35 x 120 x 120
C = 768                                      # ViT-B/16 feature width = embed_dims everywhere
voxel_size = [0.2, 0.2, 8]

cross_attention = dict(
    type='SpatialCrossAttention', pc_range=pc_range, embed_dims=C,
    deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=C, num_points=8, num_levels=1))

encoder = dict(
    type='VoxelFormerEncoder', num_layers=3, pc_range=pc_range, num_points_in_voxel=4, return_intermediate=False,
    transformerlayers=dict(type='VoxelFormerLayer', attn_cfgs=[cross_attention], feedforward_channels=2 * C,
                           ffn_dropout=0.1, operation_order=('cross_attn', 'norm', 'ffn', 'norm')))

decoder = dict(
    type='VoxelDetectionTransformerDecoder', num_layers=6, return_intermediate=True,
    transformerlayers=dict(
        type='DetrTransformerDecoderLayer',
        attn_cfgs=[dict(type='MultiheadAttention', embed_dims=C, num_heads=8, dropout=0.1),
                   dict(type='VoxelCustomMSDeformableAttention', embed_dims=C, num_levels=1)],
        ffn_cfgs=dict(type='FFN', embed_dims=768, feedforward_channels=1024, num_fcs=2, ffn_drop=0.,
                      act_cfg=dict(type='ReLU', inplace=True)),
        feedforward_channels=2 * C, ffn_dropout=0.1,
        operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))

focal = dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25)

model = dict(
    type='VoxelFormer',
    pts_bbox_head=dict(
        type='VoxelFormerOccupancyHead', bev_h=grid['h'], bev_w=grid['w'], bev_z=grid['z'], getbev=None,
        num_query=100, num_classes=17, in_channels=C, sync_cls_avg_factor=True, with_box_refine=True,
        as_two_stage=False, point_cloud_range=pc_range, occupancy_size=[0.1, 0.1, 0.1], occ_dims=128,
        occupancy_classes=16, only_occ=False, only_det=False, refine_occ=True,
        transformer=dict(type='VoxelPerceptionTransformer', rotate_prev_bev=True, use_shift=True, use_can_bus=True,
                         embed_dims=C, decoder_on_bev=False, encoder=encoder, decoder=decoder),
        bbox_coder=dict(type='NMSFreeCoder', post_center_range=[-10, -10, -5.0, 10, 10, 5.0], pc_range=pc_range,
                        max_num=50, voxel_size=voxel_size, num_classes=17),
        positional_encoding=dict(type='VoxelLearnedPositionalEncoding', num_feats=C // 2, row_num_embed=grid['h'],
                                 col_num_embed=grid['w'], z_num_embed=grid['z']),
        loss_cls=dict(focal, loss_weight=2.0),
        loss_bbox=dict(type='L1Loss', loss_weight=0.25),
        loss_iou=dict(type='GIoULoss', loss_weight=0.0),
        loss_occupancy=dict(focal, loss_weight=1.0)),
    train_cfg=dict(pts=dict(
        grid_size=[512, 512, 1], voxel_size=voxel_size, point_cloud_range=pc_range, out_size_factor=4,
        assigner=dict(type='HungarianAssigner3D', cls_cost=dict(type='FocalLossCost', weight=2.0),
                      reg_cost=dict(type='BBox3DL1Cost', weight=0.25), iou_cost=dict(type='IoUCost', weight=0.0),
                      pc_range=pc_range))))

data = dict(samples_per_gpu=1)               # the reference's viewpoints per GPU and step (bench.py: config.latency)
optimizer = dict(type='AdamW', lr=1e-4, weight_decay=0.01)
optimizer_config = dict(grad_clip=dict(max_norm=300, norm_type=2))
